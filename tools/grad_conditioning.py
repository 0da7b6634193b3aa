"""CPU experiment: how well conditioned are the parameter gradients of a full-size fixture?  The oracle (validated against the reference on
the same fixture, tests/test_oracle_golden.py) run twice, the second time with ONE-ulp moves in 1 % of the entries of an input -- the
density factors, or the ray directions -- i.e. the size of perturbation that any other implementation of the same arithmetic (another
libm, another summation order: the GPU) introduces everywhere.  Prints the relative L2 change of every parameter gradient.
    python tools/grad_conditioning.py e2e_g300_steady_1k [density|rays]       (about two minutes on 4 cores)
Round 6: on `e2e_g300_steady_1k` (1024 rays at 300^3) the density factors' gradients move by tens of per cent under such a perturbation
(the orientation term differentiates normalize(grad sigma) at samples inside the solid, where |grad sigma| is round-off), while every other
gradient moves by 1e-6 ... 1e-3: the tolerance of that fixture's density gradients in tests/test_hip_e2e.py / test_hip_timed_path.py."""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from conftest import Golden  # noqa: E402
from nmf_amd import synthetic  # noqa: E402
from oracle import nmf_oracle as O  # noqa: E402

torch.set_num_threads(int(os.environ.get("THREADS", "4")))
g = Golden(sys.argv[1])
what = sys.argv[2] if len(sys.argv) > 2 else "density"
G, BG, B = g["grid"], g["bg_res"], g["n_rays"]


def ulp_moves(t, gen, frac=float(os.environ.get("FRAC", "0.01"))):
    m = torch.rand(t.shape, generator=gen) < frac
    up = torch.nextafter(t, torch.full_like(t, float("inf")))
    return torch.where(m, up, t)


def run(perturb):
    sd = synthetic.state_dict_s1(grid=G, bg_resolution=BG, seed=0)
    gen = torch.Generator().manual_seed(1)
    if perturb and what == "density":
        for k in list(sd):
            if k.startswith("rf.density_rf."):
                sd[k] = ulp_moves(sd[k], gen)
    for k, v in sd.items():
        if k != "model.brdf_sampler.angs":
            v.requires_grad_(True)
    cfg = O.Cfg(grid=G, detach_N=False, max_retrace_rays=(g["max_retrace"],))
    vol = O.dense_alpha_mask({k: v.detach() for k, v in sd.items()}, cfg)
    rays, focal = synthetic.camera_rays(B, seed=g["ray_seed"])
    if perturb and what == "rays":
        rays = torch.cat([rays[:, :3], ulp_moves(rays[:, 3:], gen)], 1)
    torch.manual_seed(g["noise_seed"])
    forced = {"retrace_order0": g["retrace_order0"]} if "retrace_order0" in g else None
    ims, st = O.render(sd, cfg, rays, focal, vol, O.Noise(draw_unused=True), is_train=True, bg_col=torch.ones(3), forced=forced)
    gt = torch.rand(B, 3, generator=torch.Generator().manual_seed(9))
    total, loss = O.training_loss(ims, st, gt, 4096, sd)
    total.backward()
    return ims["rgb_map"].detach(), list(st["n_samples"]), {k: v.grad.clone() for k, v in sd.items() if v.grad is not None}


r0, n0, g0 = run(False)
r1, n1, g1 = run(True)
print(f"{sys.argv[1]}: one-ulp moves in {100 * float(os.environ.get('FRAC', '0.01')):.0f} % of the {what} entries; n_samples {n0} -> {n1}")
d = (r0 - r1).abs() if r0.shape == r1.shape else None
if d is not None:
    print("rgb_map: max |d|", float(d.max()), " mean", float(d.mean()))
for k in sorted(g0):
    a, b = g0[k].double(), g1[k].double()
    if float(a.norm()) > 0:
        print(f"{k:52s} rel L2 {float((a - b).norm() / a.norm()):.2e}   worst element / max {float((a - b).abs().max() / a.abs().max()):.2e}")
