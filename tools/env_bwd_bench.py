"""Isolated timing of the environment-map lookup backward on the lookups of a real training step: the arguments of every
hip.sat_lookup_bwd call of one bench step are captured and replayed (HIP events, nothing else on the device).

    python tools/env_bwd_bench.py [--reps 20]
Prints per captured call: lookups, direct scatter us, binned us, binned without the direction adjoint us; the two halves of the
binned call on their own (nmf_sat_lookup_bwd_dirs / the table role without riders) and the per-kernel times (launch probe)."""
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
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--modes", default="direct,binned,binned_no_dirs", help="under rocprofv3: one mode, so that the kernel stats are its own")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    nerf, params = bench.build(dev, grid=a.grid)
    nerf.fused_training_pass = False        # the operator graph: its calls go through hip.sat_lookup_bwd, where the spy sits
    tr = Trainer(nerf, params, tape_free=False)
    batches, focal = bench.make_batches(nerf, 3, bench.CHUNK, 0, dev)
    noise = DeviceNoise(dev, seed=1)
    for i in range(2):
        tr.step(*batches[i], focal, noise=noise, update_controllers=False, fixed_chunk=bench.CHUNK)
    calls = []
    orig = hip.sat_lookup_bwd

    def spy(sat, dirs, sa, mipbias, d_out, d_sat, d_pole, d_mip=None, want_dirs=True, want_mipbias=None, sc=None):
        calls.append((sat, dirs.clone(), sa.clone(), mipbias, d_out.clone(), sc))
        return orig(sat, dirs, sa, mipbias, d_out, d_sat, d_pole, d_mip, want_dirs, want_mipbias, sc)

    hip.sat_lookup_bwd = spy
    tr.step(*batches[2], focal, noise=noise, update_controllers=False, fixed_chunk=bench.CHUNK)
    hip.sat_lookup_bwd = orig
    torch.cuda.synchronize()
    H, W = nerf.bg_module.hw()

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

    for sat, dirs, sa, mipbias, d_out, sc in calls:
        d_sat, d_pole, d_mip = torch.zeros(H, W, 4, device=dev), torch.zeros(2, 3, device=dev), torch.zeros(1, device=dev)
        out = {}
        for name, thr, wd in (("direct", 1 << 62, True), ("binned", 1, True), ("binned_no_dirs", 1, False)):
            if name not in a.modes.split(","):
                out[name] = float("nan")
                continue
            hip.ENV_BINNED_MIN_LOOKUPS = thr
            out[name] = timed(lambda: orig(sat, dirs, sa, mipbias, d_out, d_sat, d_pole, d_mip if wd else None, wd, None, sc))
        fx = hip.HOST_EXT
        st = torch.cuda.current_stream().cuda_stream
        go = d_out.contiguous()
        t_dirs = timed(lambda: fx.sat_lookup_bwd_dirs(sat, dirs, sa, float(mipbias), go, d_pole, d_mip, sc, st))
        t_table = timed(lambda: fx.sat_lookup_bwd_table(sat, dirs, sa, float(mipbias), go, d_sat, sc, st))
        per = {}
        for name, fn in (("binned", lambda: orig(sat, dirs, sa, mipbias, d_out, d_sat, d_pole, d_mip, True, None, sc)),
                         ("table", lambda: fx.sat_lookup_bwd_table(sat, dirs, sa, float(mipbias), go, d_sat, sc, st))):
            hip.ENV_BINNED_MIN_LOOKUPS = 1
            torch.cuda.synchronize()
            fx.kernel_timing_begin("", False)
            for _ in range(a.reps):
                fn()
            t = fx.kernel_timing_end()
            per[name] = "  ".join(f"{k.replace('k_env_bin_', '')} {1e3 * v[0] / a.reps:.1f}" for k, v in sorted(t.items()) if k.startswith("k_env"))
        torch.cuda.synchronize()
        print(f"   the halves (R5): dirs role alone {t_dirs:7.1f} us   table role alone {t_table:7.1f} us")
        print(f"   per kernel, one call: {per['binned']}   | table role alone: {per['table']}")
        print(f"lookups {dirs.shape[0]:7d}  sa mean {float(sa.mean()):6.2f}  direct {out['direct']:7.1f} us  binned {out['binned']:7.1f} us  "
              f"binned without d_dirs {out['binned_no_dirs']:7.1f} us")


if __name__ == "__main__":
    main()
