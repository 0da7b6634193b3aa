"""Container-only harness that imports the *reference* (read-only, /root/reference) so that
golden vectors can be generated from it.  Nothing in here travels to the GPU box at run time:
the `-m gpu` tests, bench.py and smoke() only read the `.npz` files this harness produced.

The reference imports 13 packages that are absent from this image; none of them carries any
arithmetic on the microfacet_tensorf2 path, so they are replaced by inert stubs.
Random draws (`torch.rand`, `rand_like`, `randn_like`) are recorded in call order so the oracle
and the HIP path can be fed the identical noise.
"""
import functools
import importlib.machinery
import sys
import types

import torch

REF = "/root/reference"


class _Anything:
    """Permissive attribute sink used for the stubbed packages."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]  # behaves as an identity decorator (warp.kernel, wp.func ...)
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()

    def __getitem__(self, k):
        return _Anything()

    def __mro_entries__(self, bases):
        return (object,)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    def _attr(attr):
        if attr.startswith("__"):             # inspect / importlib probe __file__, __wrapped__, ...: behave like a plain module
            raise AttributeError(attr)
        return _Anything()

    m.__getattr__ = _attr
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    if "icecream" in sys.modules and getattr(sys.modules["icecream"], "_nmf_stub", False):
        return

    def ic(*a):
        return a[0] if len(a) == 1 else a

    _stub("icecream", ic=ic, _nmf_stub=True)

    class _Logger:
        def __getattr__(self, n):
            return lambda *a, **k: None

    _stub("loguru", logger=_Logger())
    for name in [
        "hydra", "hydra.utils", "omegaconf", "cv2", "imageio", "torchvision",
        "torchvision.transforms", "plyfile", "skimage", "skimage.measure", "lpips", "warp",
        "kornia", "trimesh", "tqdm", "tqdm.auto", "plotly", "plotly.express",
        "plotly.graph_objects", "matplotlib", "matplotlib.pyplot", "sklearn", "sklearn.linear_model",
    ]:
        if name in ("tqdm", "tqdm.auto", "matplotlib", "matplotlib.pyplot", "sklearn",
                    "sklearn.linear_model", "plotly", "plotly.express", "plotly.graph_objects"):
            try:
                __import__(name)
                continue
            except Exception:
                pass
        _stub(name)
    if REF not in sys.path:
        sys.path.insert(0, REF)


class NoiseTape:
    """Records every torch.rand / rand_like / randn_like result in call order."""

    def __init__(self):
        self.draws = []
        self._orig = {}

    def __enter__(self):
        tape = self

        def wrap(fn, kind):
            @functools.wraps(fn)
            def inner(*a, **k):
                out = fn(*a, **k)
                tape.draws.append((kind, out.detach().clone()))
                return out
            return inner

        for name in ("rand", "rand_like", "randn_like"):
            self._orig[name] = getattr(torch, name)
            setattr(torch, name, wrap(self._orig[name], name))
        return self

    def __exit__(self, *exc):
        for name, fn in self._orig.items():
            setattr(torch, name, fn)
        return False


# --- resolved microfacet_tensorf2 + tensorf_og values (configs/model/microfacet_tensorf2.yaml,
#     configs/field/tensorf_og.yaml of the reference; restated as plain kwargs) -----------------
def field_kwargs(grid):
    return dict(
        distance_scale=25, density_n_comp=16, appearance_n_comp=24, app_dim=24, step_ratio=0.5,
        density_res_multi=1, contract_space=False, smoothing=1, activation="softplus",
        interp_mode="bilinear", init_mode="rand", d_init_val=0.1, app_init_val=0.1,
        density_shift=-4, dbasis=False, grid_size=[grid] * 3, N_voxel_init=2097156,
        N_voxel_final=27000000, upsamp_list=[2000, 3000, 4000, 5500, 7000], lr=2e-2, lr_net=1e-3,
        triplanar=False, num_pretrain=0, calibrate=False,
    )


def build_reference(grid=128, near_far=(2.5, 7.0), bg_resolution=512, aabb_half=1.5,
                    max_samples=200000, max_brdf_rays=(650000, 450000), max_retrace_rays=(1000,),
                    target_num_samples=(1000000,), seed=0):
    """Hand-instantiates the reference TensorNeRF exactly as hydra would."""
    install_stubs()
    from brdf_samplers.ggx import GGXSampler
    from fields.tensoRF import TensorVMSplit
    from models.microfacet import Microfacet
    from modules.brdf import MLPBRDF
    from modules.integral_equirect import IntegralEquirect
    from modules.ish import ListISH
    from modules.render_modules import RandHydraMLPDiffuse
    from modules.tensor_nerf import TensorNeRF
    from modules.tonemap import SRGBTonemap
    from samplers.alphagrid import AlphaGridSampler

    P = functools.partial
    torch.manual_seed(seed)
    aabb = torch.tensor([[-aabb_half] * 3, [aabb_half] * 3])
    nerf = TensorNeRF(
        rf=P(TensorVMSplit, **field_kwargs(grid)),
        model=P(
            Microfacet,
            percent_bright=0.0, min_rough_start=0.0, min_rough_decay=0.999,
            max_brdf_rays=list(max_brdf_rays), conserve_energy=True,
            target_num_samples=list(target_num_samples), russian_roulette=False,
            max_retrace_rays=list(max_retrace_rays), start_std=0.0, std_decay=1.0,
            cold_start_bg_iters=0, detach_N_iters=0, anoise=0.25, no_emitters=True,
            diffuse_mixing_mode="fresnel", freeze=False, rays_per_ray=128, test_rays_per_ray=128,
            brdf_sampler=P(GGXSampler),
            brdf=P(MLPBRDF, mul_LdotN=False, feape=0, dotpe=-1,
                   h_encoder=ListISH(degs=[0, 1, 2, 4]), d_encoder=ListISH(degs=[0, 1, 2, 4]),
                   hidden_w=64, num_layers=3, initializer="kaiming", bias=0, activation="sigmoid",
                   lr=1e-3),
            diffuse_module=P(RandHydraMLPDiffuse, pospe=-1, feape=0, roughness_view_encoder=None,
                             roughness_cfg=dict(hidden_w=64, num_layers=1), hidden_w=64,
                             num_layers=1, initializer="xavier_sigmoid", lr=1e-3,
                             start_roughness=0.35, tint_bias=0, diffuse_bias=-0.619,
                             diffuse_mul=1.5, roughness_bias=-1),
            visibility_module=None,
        ),
        aabb=aabb, near_far=list(near_far),
        sampler=P(AlphaGridSampler, enable_alpha_mask=True,
                  update_list=[2000, 3000, 4000, 5500, 7000], max_samples=max_samples),
        tonemap=SRGBTonemap(),
        bg_module=IntegralEquirect(bg_resolution=bg_resolution, mipbias=1, activation="exp", lr=0.02,
                                   init_val=-0.6, mul_lr=0, brightness_lr=0, betas=[0.9, 0.99],
                                   mul_betas=[0.9, 0.9], mipbias_lr=1e-4, mipnoise=0.0),
        recur_alpha_thres=1e-3, lr_scale=1, infinity_border=False, eval_batch_size=4096,
        recur_stepmul=0.5, hdr=False, bg_noise=0.0, bg_noise_decay=0.999,
        use_predicted_normals=False, orient_world_normals=True, align_pred_norms=True,
        detach_inter=False, geonorm_iters=-1, geonorm_interp_iters=1000, contraction="AABB",
    )
    nerf.train()
    nerf.sampler.update(nerf.rf, init=True)
    return nerf
