"""Where does a one-ulp move of the density factors go?  The field query of scene S1 (values, density gradient, normals) at sample
positions inside the solid, at its surface and outside, with the factors as they are and with every entry moved by one ulp: the HIP
kernels (nmf_vm_pack_density + nmf_vm_query_fwd) against the CPU oracle's arithmetic on the same positions.  Prints, per region, the
median and the 99th percentile of |delta g| / |g| and of the angle between the normals.  Behind tools/grad_conditioning*.py: which of the
two implementations amplifies a last-bit change of its inputs, and where.
    python tools/field_sensitivity.py [grid]"""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from nmf_amd import synthetic  # noqa: E402
from oracle import nmf_oracle as O  # noqa: E402
import test_hip_parity as P  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 128
hip = P._hip()
cfg = O.Cfg(grid=G)
gen = torch.Generator().manual_seed(0)
n = 60000
regions = {"inside (|x| < 0.6)": (torch.rand(n, 3, generator=gen) * 2 - 1) * 0.6,
           "surface shell (0.70 < max|x| < 0.80)": None, "outside (|x| > 0.9)": None}
x = (torch.rand(4 * n, 3, generator=gen) * 2 - 1) * 0.8
m = x.abs().max(1).values
regions["surface shell (0.70 < max|x| < 0.80)"] = x[(m > 0.70)][:n]
x = (torch.rand(4 * n, 3, generator=gen) * 2 - 1) * 1.4
regions["outside (|x| > 0.9)"] = x[x.abs().max(1).values > 0.9][:n]


def tables(perturb):
    sd = synthetic.state_dict_s1(grid=G, bg_resolution=32, seed=0)
    if perturb:
        for k in list(sd):
            if k.startswith("rf.density_rf."):
                sd[k] = torch.nextafter(sd[k], torch.full_like(sd[k], float("inf")))
    return sd


def query(sd, xyz):
    xyz4 = torch.cat([xyz, torch.zeros(xyz.shape[0], 1)], 1)
    g_o = O.density_gradient(sd, cfg, xyz4).detach()
    n_o = O.normals(sd, cfg, xyz4).detach()
    tabs = P._field_tables(hip, sd, cfg)
    sf, sg, gr, nr, ap, cf = hip.vm_query_fwd(tabs[0], xyz4.to(P.DEV).contiguous(), *tabs[1:], want_coef=True)
    return g_o, n_o, gr.cpu(), nr.cpu()


a, b = tables(False), tables(True)
print(f"scene S1 at {G}^3; one-ulp move of every density factor entry")
print(f"{'region':40s} {'|g| median':>11s} | {'oracle d|g|/|g| med':>20s} {'p99':>9s} {'angle p99':>10s} | {'HIP d|g|/|g| med':>17s} {'p99':>9s} {'angle p99':>10s} | {'HIP vs oracle |dg|/|g| med':>26s} {'p99':>9s}")
for name, xyz in regions.items():
    go0, no0, gh0, nh0 = query(a, xyz)
    go1, no1, gh1, nh1 = query(b, xyz)
    ng = go0.norm(dim=-1).clip(min=1e-30)
    q = lambda t, p: float(torch.quantile(t.double(), p))  # noqa: E731
    ro = (go1 - go0).norm(dim=-1) / ng
    rh = (gh1 - gh0).norm(dim=-1) / ng
    ao = torch.acos((no0 * no1).sum(-1).clip(-1, 1))
    ah = torch.acos((nh0 * nh1).sum(-1).clip(-1, 1))
    rx = (gh0 - go0).norm(dim=-1) / ng
    print(f"{name:40s} {q(ng, 0.5):11.3e} | {q(ro, 0.5):20.2e} {q(ro, 0.99):9.2e} {q(ao, 0.99):10.2e} | {q(rh, 0.5):17.2e} {q(rh, 0.99):9.2e} "
          f"{q(ah, 0.99):10.2e} | {q(rx, 0.5):26.2e} {q(rx, 0.99):9.2e}")
