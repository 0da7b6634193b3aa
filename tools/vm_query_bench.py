"""Field query variants on ray-ordered samples at 128^3 and 300^3 (docs/DESIGN_rounds_1-5.md section 0.1): lane per sample (k_vm_fwd, density + normal), 16 lanes per
sample (k_vm_rows_dn, same bits), and the value-only query on the packed tables vs on the density factors themselves.

    python tools/vm_query_bench.py
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmf_amd import hip
from nmf_amd.config import build_model
dev = "cuda"
for G in (128, 300):
    nerf, _ = build_model(grid=G, bg_resolution=64, device=dev)
    rf = nerf.rf
    p, dpk, dlk, apl, ali, basis = rf._tables()
    gen = torch.Generator().manual_seed(1)
    for M in (178000, 840000):
        # samples along rays: consecutive samples are close (as in a step)
        o = (torch.rand(M // 40 + 1, 1, 3, generator=gen) * 2 - 1) * 1.2
        d = torch.nn.functional.normalize(torch.randn(M // 40 + 1, 1, 3, generator=gen), dim=-1)
        t = torch.arange(40).reshape(1, 40, 1) * (3.0 / G)
        xyz = (o + d * t).reshape(-1, 3)[:M]
        xyzt = torch.cat([xyz, torch.zeros(M, 1)], 1).to(dev).contiguous()
        def tm(fn, n=20):
            for _ in range(3): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(n): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n * 1e3
        a = tm(lambda: hip.vm_query_fwd(p, xyzt, dpk, dlk, apl, ali, basis, want_app=False))
        b = tm(lambda: hip.vm_query_rows(p, xyzt, dpk, dlk))
        c = tm(lambda: hip.vm_query_fwd(p, xyzt, dpk, dlk, apl, ali, basis, want_normal=False, want_app=False))
        vt = rf._value_tables()
        e = tm(lambda: hip.vm_query_sigma(p, xyzt, vt[0], vt[1]))
        r1 = hip.vm_query_fwd(p, xyzt, dpk, dlk, apl, ali, basis, want_app=False); r2 = hip.vm_query_rows(p, xyzt, dpk, dlk)
        same = torch.equal(r1[0], r2[0]) and torch.equal(r1[3], r2[2])
        print(f"G={G} M={M}: lane-per-sample density+normal {a:.1f} us | 16 lanes per sample {b:.1f} us (same bits: {same}) | value only: packed {c:.1f} us, factors {e:.1f} us")
