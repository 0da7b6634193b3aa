"""CPU experiment behind the tolerances of the high-specular fixture (tests/golden/e2e_variant_steady.npz): the oracle run twice on it,
once with 1 % of the env-map activations moved by one ulp (what a GPU expf does to the summed-area table): which gradients move, by how much.
    python tools/sat_sensitivity.py e2e_variant_steady      (about a minute on 8 cores)"""
import sys, torch, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from conftest import Golden
from nmf_amd import synthetic
from oracle import nmf_oracle as O
torch.set_num_threads(8)
g = Golden(sys.argv[1])
G, BG, B = g["grid"], g["bg_res"], g["n_rays"]
extra = {}
if "near_far" in g:
    h = float(g["aabb_half"])
    extra = dict(near_far=tuple(float(v) for v in g.np("near_far")), aabb=torch.tensor([[-h] * 3, [h] * 3]), roughness_bias=float(g["roughness_bias"]))
orig_act = O.env_activation
def run(perturb):
    sd = synthetic.state_dict_s1(grid=G, bg_resolution=BG, seed=0)
    for k, v in sd.items():
        if k != "model.brdf_sampler.angs": v.requires_grad_(True)
    cfg = O.Cfg(grid=G, detach_N=False, max_retrace_rays=(g["max_retrace"],), **extra)
    vol = O.dense_alpha_mask({k: v.detach() for k, v in sd.items()}, cfg)
    if perturb:
        gen = torch.Generator().manual_seed(1)
        def act(sd_):
            a = orig_act(sd_)
            m = (torch.rand(a.shape, generator=gen) < 0.01)
            up = torch.nextafter(a.detach(), torch.full_like(a, float("inf"))) - a.detach()
            return a + (up * m)
        O.env_activation = act
    else:
        O.env_activation = orig_act
    kw = dict(eye=tuple(float(v) for v in g.np("eye"))) if "eye" in g else {}
    rays, focal = synthetic.camera_rays(B, seed=g["ray_seed"], **kw)
    torch.manual_seed(g["noise_seed"])
    forced = {"retrace_order0": g["retrace_order0"]} if "retrace_order0" in g else None
    ims, st = O.render(sd, cfg, rays, focal, vol, O.Noise(draw_unused=True), is_train=True, bg_col=torch.ones(3), forced=forced)
    gt = torch.rand(B, 3, generator=torch.Generator().manual_seed(9))
    total, loss = O.training_loss(ims, st, gt, 4096, sd)
    total.backward()
    return ims["rgb_map"].detach(), {k: v.grad.clone() for k, v in sd.items() if v.grad is not None}
r0, g0 = run(False)
r1, g1 = run(True)
d = (r0 - r1).abs()
print("rgb: max", float(d.max()), "mean", float(d.mean()), "frac>1e-4", float((d > 1e-4).float().mean()))
for k in ("rf.density_rf.app_plane.0", "rf.density_rf.app_plane.1", "rf.density_rf.app_line.0", "model.diffuse_module.roughness_mlp.0.weight", "bg_module.mipbias", "model.brdf.mlp.0.weight", "bg_module.bg_mat"):
    a, b = g0[k].double(), g1[k].double()
    print(f"{k:48s} rel L2 {float((a-b).norm()/a.norm()):.2e}")
