"""Isolated timing of nmf_composite_fwd / _bwd on the segments of a real training step (captured from the operator graph of
nmf_amd/functional.py, whose calls go through hip.composite_*).  (Round 4 compared 8 against 16 lanes per ray with this tool through an
environment knob of the library; the result is fixed in csrc/composite.hip: forward 8, backward 16.)

    python tools/composite_bench.py [--reps 20]"""
import argparse
import os
import sys

import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from nmf_amd import hip  # noqa: E402
from nmf_amd.noise import DeviceNoise  # noqa: E402
from nmf_amd.trainer import Trainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    nerf, params = bench.build(dev)
    nerf.fused_training_pass = False        # the operator graph: its calls go through hip.composite_*, where the spy sits
    tr = Trainer(nerf, params, tape_free=False)
    batches, focal = bench.make_batches(nerf, 3, bench.CHUNK, 0, dev)
    noise = DeviceNoise(dev, seed=1)
    for i in range(2):
        tr.step(*batches[i], focal, noise=noise, update_controllers=False, fixed_chunk=bench.CHUNK)
    calls = []
    orig = hip.composite_bwd

    def spy(sigma, dist, weight, offsets, b, scale, d_weight):
        calls.append((sigma.clone(), dist.clone(), weight.clone(), offsets.clone(), int(b), float(scale), d_weight.clone()))
        return orig(sigma, dist, weight, offsets, b, scale, d_weight)

    hip.composite_bwd = spy
    tr.step(*batches[2], focal, noise=noise, update_controllers=False, fixed_chunk=bench.CHUNK)
    hip.composite_bwd = orig
    torch.cuda.synchronize()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(a.reps):
            fn()
        e.record()
        e.synchronize()
        return s.elapsed_time(e) / a.reps * 1e3

    for sigma, dist, weight, offsets, b, scale, d_weight in calls:
        n = (offsets[1:b + 1] - offsets[:b]).cpu()
        q = [int((n > k).sum()) for k in (0, 8, 16, 32, 64)]
        print(f"rays {b}  samples {sigma.shape[0]}  mean {float(n.float().mean()):.2f}  max {int(n.max())}  rays with > 0 / 8 / 16 / 32 / 64 samples: {q}")
        res = {}
        for W in ("8",):
            res[W] = (orig(sigma, dist, weight, offsets, b, scale, d_weight), hip.composite_fwd(sigma, dist, offsets, b, scale))
            tb = timed(lambda: orig(sigma, dist, weight, offsets, b, scale, d_weight))
            tf = timed(lambda: hip.composite_fwd(sigma, dist, offsets, b, scale))
            print(f"   W = {W:2s}: fwd {tf:6.1f} us   bwd {tb:6.1f} us")


if __name__ == "__main__":
    main()
