"""Training entry point -- counterpart of the reference's train.py:191-901 for `model=microfacet_tensorf2` on the
synthetic "S2 orbit" data set (nerf_synthetic is not available offline): ground-truth images are
rendered from the S1 scene itself (eval mode), then a freshly initialised model is fitted to them.

    python -m nmf_amd.train model=microfacet_tensorf2 field=tensorf_og dataset=lego datadir=/data expname=lego [a.b.c=value ...]
    python -m nmf_amd.train -m expname=v38 model=microfacet_tensorf2 dataset=ficus,drums,ship datadir=/data     (hydra multirun, README.md:10)
    python -m nmf_amd.train --iters 200 --views 24 --res 64 [--grid 64] [--eval-every 100]
    python -m nmf_amd.train --datadir /data/nerf_synthetic/lego --near-far 2.5 7 --iters 30000 --grid 128 --bg 512
    python -m torch.distributed.run --nproc-per-node N -m nmf_amd.train ...        (data parallel, RCCL)

The first form is the reference's hydra command line (train.py:904-921): `group=name` choices and dotted overrides are composed by
nmf_amd/yaml_config.py -- from the built-in tree of nmf_amd/config.py (the values of configs/default.yaml, model/
microfacet_tensorf2.yaml, field/tensorf_og.yaml, dataset/<scene>.yaml) or, with --config-dir, from a configs/ directory such as
the reference's --, the model is built by `instantiate_arch` from the composed `model.arch` (`_target_` / `_partial_`,
train.py:239-247), the trainer reads `model.params`, and the resolved config is written to <basedir>/<scene>_<expname>/config.yaml
(train.py:193,226,485).  `-m` / `--multirun`: every comma-separated override value spans an axis and the runs of the product are carried
out one after the other in this process (hydra's basic sweeper); each writes its own <scene>_<expname> folder.  The flags of the other forms are shorthands for overrides of the same tree (--grid = field.grid_size, --bg =
model.arch.bg_module.bg_resolution, ...); dataset=s2_orbit (the default without a data directory) is the offline stand-in.

With --datadir the rays and colours come from a Blender / nerf_synthetic scene directory (nmf_amd/dataLoader/blender.py,
RGBA frames blended onto the white background as train.py:525-530 does); --save writes a checkpoint readable by
TensorNeRF.load.

Prints one JSON line per evaluation: iteration, train PSNR proxy (train.py:609-613), test PSNR with the reference's
8-bit formula (renderer.py:399-401), rays/s.
"""
import argparse
import json
import os
import time

import torch

from . import synthetic
from . import yaml_config
from .noise import DeviceNoise
from .renderer import psnr_8bit, render_images as _render_images
from .trainer import Trainer, agree, rank_slice


def render_images(nerf, rays, focal, chunk, noise):
    """chunk_renderer with render2completion (renderer.py:56-106) in eval mode -> rgb [n,3]"""
    return _render_images(nerf, rays, focal, chunk, noise, draw_debug=True)


def compose_run(args, overrides):
    """argparse flags + hydra-style tokens -> the resolved config of the run (flags are shorthands for overrides)"""
    ov = []
    if args.datadir:
        # a scene directory given directly: the dataset entry points at it (the reference splits it into datadir / scenedir; the two
        # paths are set on the composed tree below -- file-system paths do not go through the YAML value parser: '007' is not 7)
        ov += ["dataset=lego"]
        if args.near_far:
            ov.append(f"dataset.near_far=[{args.near_far[0]},{args.near_far[1]}]")
        if args.downsample != 1.0:
            ov += [f"dataset.downsample_train={args.downsample}", f"dataset.downsample_test={args.downsample}"]
    elif not any(o.split("=")[0] in ("dataset", "datadir") for o in overrides):
        ov += ["dataset=s2_orbit"]
    if args.grid is not None:
        ov.append(f"field.grid_size=[{args.grid},{args.grid},{args.grid}]")
    if args.bg is not None:
        ov.append(f"model.arch.bg_module.bg_resolution={args.bg}")
    if args.seed is not None:
        ov.append(f"seed={args.seed}")
    if args.views is not None:
        ov.append(f"dataset.views={args.views}")
    if args.test_views is not None:
        ov.append(f"N_vis={args.test_views}")
    if args.res is not None:
        ov.append(f"dataset.res={args.res}")
    cfg = yaml_config.compose(args.config_dir, ov + list(overrides))
    if args.datadir:
        path = os.path.abspath(args.datadir)
        cfg["datadir"], cfg["dataset"]["scenedir"] = os.path.dirname(path) or "/", os.path.basename(path)
    return cfg


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config-dir", type=str, default=None,
                    help="configs/ directory to compose from (e.g. the reference's); default: the built-in tree (nmf_amd/config.py)")
    ap.add_argument("--iters", type=int, default=None,
                    help="optimizer steps of this run (default: model.params.n_iters; 200 when only flags are given)")
    ap.add_argument("--views", type=int, default=None)
    ap.add_argument("--test-views", type=int, default=None)
    ap.add_argument("--res", type=int, default=None)
    ap.add_argument("--grid", type=int, default=None)
    ap.add_argument("--bg", type=int, default=None)
    ap.add_argument("--eval-every", type=int, default=None, help="default: vis_every of the config")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--datadir", type=str, default=None, help="Blender scene directory (transforms_*.json + frames)")
    ap.add_argument("--near-far", type=float, nargs=2, default=None)
    ap.add_argument("--downsample", type=float, default=1.0)
    ap.add_argument("--save", type=str, default=None, help="write a checkpoint (TensorNeRF.save) at the end")
    ap.add_argument("--table-dtype", choices=("f32", "bf16"), default="f32",
                    help="bf16: the forward field queries read bfloat16 copies of the factor tables (BASELINE configs[1])")
    ap.add_argument("--rays-per-gpu", type=int, default=None,
                    help="weak scaling: every rank takes this many rays per optimizer step (BASELINE configs[3]: 32768), "
                         "processed in num_rays chunks; default: the reference's lbatch_size split over the ranks")
    ap.add_argument("--no-config-file", action="store_true", help="do not write <basedir>/<expname>/config.yaml")
    ap.add_argument("-m", "--multirun", action="store_true",
                    help="hydra multirun (README.md:10): comma-separated override values span a sweep, run job by job")
    ap.add_argument("overrides", nargs="*", help="hydra-style tokens: group=name, a.b.c=value")
    args = ap.parse_args(argv)
    for o in args.overrides:
        if "=" not in o:
            ap.error(f"'{o}': overrides are key=value tokens (hydra syntax)")
    if args.multirun:
        # the product of the comma-separated values, in hydra's order (the last axis varies fastest); one job after the other
        jobs = [combo for combo, _cfg in yaml_config.sweep(args.config_dir, args.overrides)]
        flags = [a for a in (argv if argv is not None else os.sys.argv[1:]) if a not in ("-m", "--multirun") and a not in args.overrides]
        done = []
        for i, combo in enumerate(jobs):
            if int(os.environ.get("RANK", "0")) == 0:
                print(json.dumps(dict(multirun_job=i, of=len(jobs), overrides=combo)), flush=True)
            done.append(main(flags + list(combo)))
        return done
    cfg = compose_run(args, args.overrides)
    shorthand = not args.overrides and args.config_dir is None       # the flag forms: small defaults (grid 64^3, env 128 x 256, 4 test views)
    if shorthand:
        if args.grid is None:
            cfg["field"]["grid_size"] = cfg["model"]["arch"]["rf"]["grid_size"] = [64, 64, 64]
        if args.bg is None:
            cfg["model"]["arch"]["bg_module"]["bg_resolution"] = 128
        if args.test_views is None and args.datadir:
            cfg["N_vis"] = 4
    ds = cfg["dataset"]
    params = cfg["model"]["params"]
    n_iters = args.iters if args.iters is not None else (200 if shorthand else int(params["n_iters"]))
    eval_every = args.eval_every if args.eval_every is not None else (100 if shorthand else int(cfg["vis_every"]))
    seed = int(cfg["seed"])
    grid = int(cfg["field"]["grid_size"][0])
    bg_res = int(cfg["model"]["arch"]["bg_module"]["bg_resolution"])

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend=os.environ.get("NMF_BACKEND", "nccl"))

    # train.py:193,226: <basedir>/<last component of the scene directory>_<expname>
    expname = f"{str(ds.get('scenedir') or ds['dataset_name']).split('/')[-1]}_{cfg['expname']}"
    logfolder = os.path.join(str(cfg["basedir"]), expname)
    if rank == 0 and not args.no_config_file and not shorthand:
        os.makedirs(logfolder, exist_ok=True)
        yaml_config.dump(cfg, os.path.join(logfolder, "config.yaml"))                     # train.py:485

    near_far = tuple(ds["near_far"])
    aabb_half = 1.5
    if ds["dataset_name"] == "blender":
        # ---- real data: Blender scene (dataLoader/blender.py), colours blended onto white (train.py:525-530)
        from .dataLoader import BlenderDataset
        scene = os.path.join(str(cfg["datadir"]), str(ds["scenedir"]))
        n_vis = int(cfg["N_vis"])
        tr_set = BlenderDataset(scene, split="train", downsample=float(ds["downsample_train"]), is_stack=False)
        te_set = BlenderDataset(scene, split="test", downsample=float(ds["downsample_test"]), is_stack=True, N_vis=n_vis)
        if args.datadir and not args.near_far and not any(o.startswith("dataset.near_far") for o in args.overrides):
            near_far = tuple(tr_set.near_far)
        rays_tr, rgba = tr_set.all_rays.to(dev), tr_set.all_rgbs.to(dev)
        rgb_tr = rgba[:, :3] * rgba[:, 3:] + (1 - rgba[:, 3:]) if rgba.shape[1] == 4 else rgba
        rays_te = te_set.all_rays.reshape(-1, 6).to(dev)
        rgb_te = te_set.all_rgbs.reshape(-1, 3).to(dev)
        focal = float(tr_set.fx)
        test_views = te_set.all_rays.shape[0]
        aabb = tr_set.scene_bbox.float() * float(ds.get("aabb_scale", 1))                # train.py:234-237
    elif ds["dataset_name"] == "synthetic_orbit":
        # ---- ground truth from the S1 scene
        torch.manual_seed(seed)
        aabb = torch.tensor([[-aabb_half] * 3, [aabb_half] * 3])
        teacher = yaml_config.instantiate_arch(cfg, aabb, near_far).to(dev)
        teacher.sampler.update(teacher.rf, init=True)
        teacher.load_state_dict(synthetic.state_dict_s1(grid=grid, bg_resolution=bg_res, seed=0), strict=False)
        teacher.eval()
        teacher.sampler.update(teacher.rf, init=False)
        teacher.sampler.update(teacher.rf, init=True)
        test_views = int(ds.get("test_views", 4)) if args.test_views is None else args.test_views
        rays_tr, focal = synthetic.orbit_rays(int(ds.get("views", 24)), int(ds.get("res", 64)), seed=1)
        rays_te, _ = synthetic.orbit_rays(test_views, int(ds.get("res", 64)), seed=2)
        rays_tr, rays_te = rays_tr.to(dev), rays_te.to(dev)
        gt_noise = DeviceNoise(dev, seed=7)
        rgb_tr = render_images(teacher, rays_tr, focal, 4096, gt_noise)
        rgb_te = render_images(teacher, rays_te, focal, 4096, gt_noise)
        del teacher
    else:
        raise NotImplementedError(f"dataset_name {ds['dataset_name']}: this package reads Blender scene directories "
                                  "(dataLoader/blender.py) and the synthetic orbit")

    # ---- student: train.py:239-247 `hydra.utils.instantiate(cfg.model.arch)(aabb=, near_far=)`, fresh initialisation
    # (SURVEY Appendix E), calibration (train.py:429-437)
    torch.manual_seed(seed)                  # identical replicas on every rank
    nerf = yaml_config.instantiate_arch(cfg, aabb, near_far).to(dev)
    nerf.sampler.update(nerf.rf, init=True)
    nerf.train()
    nerf.rf.set_table_dtype(args.table_dtype)
    with torch.no_grad():
        xyz = torch.rand(100000, 4, device=dev) * 2 - 1
        xyz[:, 3] *= 0
        feat = nerf.rf.compute_appfeature(xyz)
        nerf.model.calibrate(None, xyz, feat, nerf.bg_module.mean_color().mean())
    if world > 1:
        # the calibrated biases come from random points (models/microfacet.py:79-96) -- rank 0's replica, biases
        # included, is what every rank trains (the seeds above make them equal already on equal devices; this makes them equal)
        from .trainer import broadcast_replica, check_replicas
        broadcast_replica(nerf, src=0)
        check_replicas(nerf, what="after the start-up broadcast")
    trainer = Trainer(nerf, params, world_size=world, rank=rank)
    noise = DeviceNoise(dev, seed=1000 + rank)
    g = torch.Generator(device=dev).manual_seed(seed)        # same permutation on every rank
    n_total = rays_tr.shape[0]
    perm, cur = torch.randperm(n_total, device=dev, generator=g), 0
    # Which rays a chunk gets.  One process: train.py:34-51 as it is (controllers.SimpleSampler: one nextids() per chunk, the cursor moved
    # before the slice -- chunks of different sizes overlap, 8 % of a steady batch are repeated rays; it decides the PSNR after equal
    # iterations).  Data parallel (the reference has none): disjoint shards of one global permutation.
    sampler = None
    if world == 1 and not args.rays_per_gpu:
        from .controllers import SimpleSampler
        sampler = SimpleSampler(n_total, int(params["batch_size"]), lambda n_: torch.randperm(n_, device=dev, generator=g))

        def fetch(n_):
            ids_ = sampler.nextids(n_)
            return rays_tr[ids_], rgb_tr[ids_]
    t0, rays_seen = time.time(), 0
    for it in range(n_iters):
        # The global batch must be the same number on every rank (it sizes the shards, advances the shared permutation and
        # normalises the loss, train.py:504-507,703) while each rank's ray controller follows its own chunks: agree on it.
        nb = world * args.rays_per_gpu if args.rays_per_gpu else agree(trainer.lbatch_size(), "min", device=dev)
        if sampler is not None:
            out = trainer.step(None, None, focal, noise=noise, global_rays=nb, fetch=fetch)
        else:
            if cur + nb > n_total:
                perm, cur = torch.randperm(n_total, device=dev, generator=g), 0
            ids = perm[cur:cur + nb][rank_slice(nb, world, rank)]      # one global permutation, sharded over ranks
            cur += nb
            out = trainer.step(rays_tr[ids], rgb_tr[ids], focal, noise=noise, global_rays=nb)
        rays_seen += out["rays"] * world
        if (it + 1) % eval_every == 0 or it + 1 == n_iters:
            nerf.eval()
            pred = render_images(nerf, rays_te, focal, 4096, noise)
            nerf.train()
            pv, gv = pred.reshape(test_views, -1, 3), rgb_te.reshape(test_views, -1, 3)
            psnr = float(torch.stack([psnr_8bit(pv[i], gv[i]) for i in range(test_views)]).mean())   # renderer.py:511-513
            if rank == 0:
                print(json.dumps(dict(iteration=it + 1, train_psnr=round(out["psnr"], 3), test_psnr=round(psnr, 3),
                                      rays_per_s=round(rays_seen / (time.time() - t0), 1), num_rays=trainer.num_rays,
                                      retrace=nerf.model.max_retrace_rays, n_samples=out["n_samples"])), flush=True)
    save = args.save
    if save is None and not shorthand and not args.no_config_file:
        save = os.path.join(logfolder, f"{expname}.th")                                     # train.py:856 (tensorf.save)
    if save and rank == 0:
        arch = cfg["model"]["arch"]
        arch["model"]["brdf"]["bias"] = nerf.model.brdf.bias                        # calibrated values (train.py:429-437)
        arch["model"]["diffuse_module"]["diffuse_bias"] = nerf.model.diffuse_module.diffuse_bias
        arch["model"]["diffuse_module"]["roughness_bias"] = nerf.model.diffuse_module.roughness_bias
        nerf.save(save, arch)
    if world > 1:
        dist.destroy_process_group()
    return cfg


if __name__ == "__main__":
    main()
