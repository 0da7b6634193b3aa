"""GPU counterpart of tools/grad_conditioning.py: the HIP path (operator graph, reference bookkeeping replayed) on a full-size fixture,
run twice -- the second time with every density factor entry moved by one ulp -- and both compared with the reference's gradients.
Says whether a deviation from the reference is the size of the path's OWN sensitivity (an ill-conditioned gradient) or beyond it.
    python tools/grad_conditioning_gpu.py e2e_g300_steady_1k"""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from conftest import Golden  # noqa: E402
import test_hip_e2e as T  # noqa: E402

name = sys.argv[1]
g = Golden(name)


def run(perturb):
    nerf = T._full_size_model(g)
    if perturb:
        with torch.no_grad():
            for k, p in nerf.named_parameters():
                if k.startswith("rf.density_rf."):
                    p.copy_(torch.nextafter(p, torch.full_like(p, float("inf"))))
    pins = T._pin_reference_bookkeeping(g)
    ims, st = T._seeded_render(nerf, g, pins)
    B = g["n_rays"]
    gt = torch.rand(B, 3, generator=torch.Generator().manual_seed(9)).to(T.DEV)
    wv = st["whole_valid"]
    loss = ((ims["rgb_map"].clip(max=1).clip(0, 1) - gt[wv].clip(0, 1)) ** 2).sum()
    total = (loss + 0.1 * st["ori_loss"] + 3e-4 * st["prediction_loss"] + 8e-5 * nerf.rf.density_L1()) / 4096
    total.backward()
    return {k: p.grad.detach().double().cpu() for k, p in nerf.named_parameters() if p.grad is not None}, list(st["n_samples"])


a, na = run(False)
a2, _ = run(False)          # the same inputs again: what the order of the float atomics alone does
b, nb = run(True)
print(f"{name}: n_samples {na} / perturbed {nb}; reference {[int(v) for v in g.np('n_samples')]}")
rel = lambda x, y: float((x - y).norm() / y.norm().clip(min=1e-30))  # noqa: E731
print(f"{'gradient':52s} {'HIP vs HIP again':>18s} {'HIP vs HIP(+1 ulp)':>20s} {'HIP vs reference':>18s} {'HIP(+1 ulp) vs ref':>20s}")
for k in g.keys("grad/") + g.keys("grad_slice4/"):
    n_ = k.split("/", 1)[1]
    pick = (lambda t: t) if k.startswith("grad/") else (lambda t: t[0, :, ::4, ::4])
    ref = torch.as_tensor(g[k]).double().reshape(pick(a[n_]).shape)
    print(f"{k:52s} {rel(pick(a[n_]), pick(a2[n_])):18.2e} {rel(pick(a[n_]), pick(b[n_])):20.2e} {rel(pick(a[n_]), ref):18.2e} {rel(pick(b[n_]), ref):20.2e}")
