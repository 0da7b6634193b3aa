"""Resolved configuration of `model=microfacet_tensorf2 field=tensorf_og` (reference:
configs/model/microfacet_tensorf2.yaml, configs/field/tensorf_og.yaml) and a builder that instantiates the
operator tree the way hydra's `_target_` / `_partial_` instantiation does in the reference (train.py:239)."""
import copy
import functools

import torch

FIELD = dict(
    _target_="fields.tensoRF.TensorVMSplit", distance_scale=25, density_n_comp=16, appearance_n_comp=24, app_dim=24,
    step_ratio=0.5, density_res_multi=1, contract_space=False, smoothing=1, activation="softplus",
    interp_mode="bilinear", init_mode="rand", d_init_val=0.1, app_init_val=0.1, density_shift=-4, dbasis=False,
    grid_size=[128, 128, 128], N_voxel_init=2097156, N_voxel_final=27000000, upsamp_list=[2000, 3000, 4000, 5500, 7000],
    lr=2e-2, lr_net=1e-3, triplanar=False, num_pretrain=0, calibrate=False)

MODEL = dict(
    arch=dict(
        _target_="modules.tensor_nerf.TensorNeRF", recur_alpha_thres=1e-3, lr_scale=1, infinity_border=False,
        eval_batch_size=4096, recur_stepmul=0.5, hdr=False, bg_noise=0.0, bg_noise_decay=0.999,
        use_predicted_normals=False, orient_world_normals=True, align_pred_norms=True, detach_inter=False,
        geonorm_iters=-1, geonorm_interp_iters=1000, contraction="AABB",
        tonemap=dict(_target_="modules.tonemap.SRGBTonemap"),
        sampler=dict(_target_="samplers.alphagrid.AlphaGridSampler", enable_alpha_mask=True,
                     update_list=[2000, 3000, 4000, 5500, 7000], max_samples=200000),
        model=dict(
            _target_="models.microfacet.Microfacet", percent_bright=0.0, min_rough_start=0.0, min_rough_decay=0.999,
            max_brdf_rays=[650000, 450000], conserve_energy=True, target_num_samples=[1000000], russian_roulette=False,
            max_retrace_rays=[1000], start_std=0.0, std_decay=1.0, cold_start_bg_iters=0, detach_N_iters=0, anoise=0.25,
            no_emitters=True, diffuse_mixing_mode="fresnel", freeze=False, rays_per_ray=128, test_rays_per_ray=128,
            brdf_sampler=dict(_target_="brdf_samplers.ggx.GGXSampler"),
            brdf=dict(_target_="modules.brdf.MLPBRDF", mul_LdotN=False, feape=0, dotpe=-1,
                      h_encoder=dict(_target_="modules.ish.ListISH", degs=[0, 1, 2, 4]),
                      d_encoder=dict(_target_="modules.ish.ListISH", degs=[0, 1, 2, 4]),
                      hidden_w=64, num_layers=3, initializer="kaiming", bias=0, activation="sigmoid", lr=1e-3),
            diffuse_module=dict(_target_="modules.render_modules.RandHydraMLPDiffuse", pospe=-1, feape=0,
                                roughness_view_encoder=None, roughness_cfg=dict(hidden_w=64, num_layers=1),
                                hidden_w=64, num_layers=1, initializer="xavier_sigmoid", lr=1e-3, start_roughness=0.35,
                                tint_bias=0, diffuse_bias=-0.619, diffuse_mul=1.5, roughness_bias=-1),
            visibility_module=None),
        bg_module=dict(_target_="modules.integral_equirect.IntegralEquirect", bg_resolution=512, mipbias=1,
                       activation="exp", lr=0.02, init_val=-0.6, mul_lr=0, brightness_lr=0, betas=[0.9, 0.99],
                       mul_betas=[0.9, 0.9], mipbias_lr=1e-4, mipnoise=0.0),
        rf="placeholder"),
    params=dict(
        L1_weight_initial=8e-5, L1_weight_rest=4e-5, clip_grad=None, weight_decay=0, eps=1e-8, betas=[0.9, 0.99],
        starting_batch_size=100, min_batch_size=4096, max_batch_size=8000, target_num_samples=200000,
        pred_lambda=3e-4, ori_lambda=0.1, n_iters=30000, batch_size=4096, lr_init=1, lr_final=1e-3, lr_delay_mult=0.1,
        lr_delay_steps=100, bg_col="white"))


# configs/default.yaml: the top-level keys of a run.  The reference's defaults list names dataset=materials, model=brdf_tcnn;
# brdf_tcnn is outside this package's path, so the built-in tree defaults to the headline configuration.
DEFAULT = dict(
    defaults=["_self_", {"dataset": "lego"}, {"model": "microfacet_tensorf2"}, {"field": "tensorf_og"}],
    basedir="./log", filter_rays=False, expname="test", datadir="/data", render_only=False, render_train=False,
    render_test=True, render_path=False, add_timestamp=False, nSamples=1e6, N_vis=5, vis_every=5000,
    progress_refresh_rate=1, rm_weight_mask_thre=1e-4, step_ratio=0.5, ckpt=None, lr_decay_iters=-1,
    lr_decay_target_ratio=0.1, lr_upsample_reset=1, fp16=False, n_bg_iters=1000, save_often=False, fixed_bg=None,
    seed=20211200, gt_bg=None, render_mode="center")

# the keys of configs/model/microfacet_tensorf2.yaml:params that Trainer does not read (they switch off terms of train.py this
# path does not have) -- carried so that a resolved config.yaml has the reference's key set
PARAMS_UNUSED = dict(
    TV_weight_density=0.0, TV_weight_app=0.0, TV_weight_bg=0, envmap_lambda=0, final_pred_lambda=None, diffuse_lambda=0,
    final_ori_lambda=None, brdf_lambda=0, normal_err_lambda=0, distortion_lambda=0, visibility_lambda=0, charbonier_eps=1e-3,
    ortho_weight=0, N_visibility_rays=128, charbonier_loss=False, start_density=1e-3, lr=None)


def _blender(scene, near_far, gt_bg=None, **extra):
    d = dict(scenedir=f"nerf_synthetic/{scene}", dataset_name="blender", downsample_train=1, downsample_test=1, ndc_ray=False,
             near_far=list(near_far))
    d.update(extra)
    if gt_bg is not None:
        d["gt_bg"] = gt_bg
    return d


# configs/dataset/<name>.yaml (values only; BASELINE.json's four scenes first)
DATASETS = dict(
    lego=_blender("lego", [2.5, 7], "lego_bg.exr"), ship=_blender("ship", [1, 6], "sunrise.exr"),
    materials=_blender("materials", [2, 6], "forest.exr", stack_norms=False),
    helmet=_blender("helmet", [3, 5], "abandoned_factory_canteen_01_4k.exr", aabb_scale=2),
    car=_blender("car", [2, 5], "forest.exr", stack_norms=False), chair=_blender("chair", [1, 6], "interior.exr"),
    drums=_blender("drums", [2.5, 6], "interior.exr"), ficus=_blender("ficus", [1, 6], "interior.exr"),
    hotdog=_blender("hotdog", [2, 5], "sunrise.exr"), mic=_blender("mic", [2, 6], "courtyard.exr"),
    toaster=_blender("toaster", [2.5, 5], "interior.exr"), coffee=_blender("coffee", [3, 5]),
    teapot=_blender("teapot", [3, 5], "sunset_jhbcentral_4k.exr", aabb_scale=0.75),
    ball=_blender("ball", [2.5, 5], "forest.exr", aabb_scale=2),
    # not a reference file: the offline stand-in of SURVEY 8(d) (views rendered from scene S1 on an orbit; no files read)
    s2_orbit=dict(scenedir=None, dataset_name="synthetic_orbit", downsample_train=1, downsample_test=1, ndc_ray=False,
                  near_far=[2.5, 7], views=24, test_views=4, res=64))


def builtin_tree():
    """The config tree `compose()` reads when no --config-dir is given: {relative path without .yaml: dict}, same layout
    as the reference's configs/ directory (default.yaml, model/<name>.yaml, field/<name>.yaml, dataset/<name>.yaml)."""
    model = copy.deepcopy(MODEL)
    order = ["L1_weight_initial", "L1_weight_rest", "clip_grad", "weight_decay", "eps", "betas", "starting_batch_size",
             "min_batch_size", "max_batch_size", "target_num_samples", "TV_weight_density", "TV_weight_app", "TV_weight_bg",
             "envmap_lambda", "pred_lambda", "final_pred_lambda", "diffuse_lambda", "ori_lambda", "final_ori_lambda",
             "brdf_lambda", "normal_err_lambda", "distortion_lambda", "visibility_lambda", "charbonier_eps", "ortho_weight",
             "N_visibility_rays", "n_iters", "charbonier_loss", "start_density", "batch_size", "lr", "lr_init", "lr_final",
             "lr_delay_mult", "lr_delay_steps", "bg_col"]
    allp = dict(PARAMS_UNUSED, **model["params"])
    model["params"] = {k: allp[k] for k in order}
    _mark_partial(model)
    tree = {"default": copy.deepcopy(DEFAULT), "model/microfacet_tensorf2": model,
            "field/tensorf_og": _mark_partial(copy.deepcopy(FIELD))}
    for name, d in DATASETS.items():
        tree[f"dataset/{name}"] = copy.deepcopy(d)
    return tree


def resolved_config():
    cfg = copy.deepcopy(MODEL)
    cfg["arch"]["rf"] = copy.deepcopy(FIELD)          # train.py:911: cfg.model.arch.rf = cfg.field
    return _mark_partial(cfg)


PARTIAL = {"fields.tensoRF.TensorVMSplit", "samplers.alphagrid.AlphaGridSampler", "models.microfacet.Microfacet",
           "brdf_samplers.ggx.GGXSampler", "modules.brdf.MLPBRDF", "modules.render_modules.RandHydraMLPDiffuse",
           "modules.tensor_nerf.TensorNeRF"}


def _mark_partial(node):
    """the nodes the reference's YAML marks `_partial_: True` (constructed later with aabb / in_channels / ...)"""
    if isinstance(node, dict):
        if node.get("_target_") in PARTIAL:
            node["_partial_"] = True
        for v in node.values():
            _mark_partial(v)
    return node


def instantiate(node):
    """hydra.utils.instantiate for the `_target_` strings this config uses (nmf_amd/yaml_config.py)."""
    from .yaml_config import instantiate as _inst
    return _inst(node)


def build_model(grid=128, bg_resolution=512, near_far=(2.5, 7.0), aabb_half=1.5, device="cuda", overrides=None):
    cfg = resolved_config()
    cfg["arch"]["rf"]["grid_size"] = [grid] * 3
    cfg["arch"]["bg_module"]["bg_resolution"] = bg_resolution
    for path, v in (overrides or {}).items():
        node = cfg["arch"]
        keys = path.split(".")
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = v
    aabb = torch.tensor([[-aabb_half] * 3, [aabb_half] * 3])
    nerf = instantiate(cfg["arch"])(aabb=aabb, near_far=list(near_far))
    nerf = nerf.to(device)
    nerf.sampler.update(nerf.rf, init=True)
    return nerf, cfg
