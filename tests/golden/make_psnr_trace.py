"""Container-only: PSNR after equal iterations at a TRAINED quality level (north_star "PSNR within
0.05 dB of reference after equal iterations").  Runs the REFERENCE's own `reconstruction()` loop (through
make_train_trace.run, nothing of the loop modified) on the S2 orbit data set for several seeds, long enough and large enough
that the reference reaches a trained image (tools/psnr_probe.py sized the configuration on the GPU: 48^3 grid, 32x32 views,
1024-ray batches, the reference's own 128 secondary rays per sample and 30000-iteration lr decay reach ~30 dB after 100 and
~35 dB after 300 iterations; 32 rays per sample never leave 11 dB -- in the reference and in this build alike), and stores per seed: the initial state dict, the calibrated biases, the CPU
generator state at the first iteration and the test PSNR per view at fixed iterations.  tests/golden/psnr_trace.npz holds
arrays only.

    python tests/golden/make_psnr_trace.py [--seeds 3] [--iters 300] ...
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_train_trace as mt  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=300, help="iterations actually run")
    ap.add_argument("--lr-iters", type=int, default=30000, help="model.params.n_iters: length of the lr decay (reference default)")
    ap.add_argument("--rpr", type=int, default=128)
    ap.add_argument("--retrace", type=int, default=1000)
    ap.add_argument("--grid0", type=int, default=48)
    ap.add_argument("--grid1", type=int, default=48)
    ap.add_argument("--upsample-at", type=int, nargs="+", default=[1000000])
    ap.add_argument("--psnr-at", type=int, nargs="+", default=[100, 200, 300])
    ap.add_argument("--res", type=int, default=32)
    ap.add_argument("--train-views", type=int, default=24)
    ap.add_argument("--test-views", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--out", default=os.path.join(HERE, "psnr_trace.npz"))
    ap.add_argument("--first-seed", type=int, default=0, help="> 0: keep the seeds already in --out and add seeds first-seed .. first-seed + seeds - 1")
    a = ap.parse_args()
    out = {}
    if a.first_seed > 0:
        with np.load(a.out) as old:
            out = {k: old[k] for k in old.files}
        assert int(out["n_seeds"]) == a.first_seed, (int(out["n_seeds"]), a.first_seed)
    for s in range(a.first_seed, a.first_seed + a.seeds):
        t0 = time.time()
        r = mt.run(grid0=a.grid0, grid1=a.grid1, teacher_grid=a.grid1, bg=32, upsample_at=tuple(a.upsample_at), n_iters=a.lr_iters, stop_at=a.iters,
                   psnr_at=tuple(a.psnr_at), res=a.res, train_views=a.train_views, test_views=a.test_views, seed=20211200 + s,
                   batch=a.batch, max_batch=2 * a.batch, max_samples=40000, max_brdf_rays=(80000, 40000),
                   target_num_samples=80000, max_retrace=a.retrace, rays_per_ray=a.rpr, light=True, threads=a.threads)
        print(f"seed {s}: {time.time() - t0:.0f} s", flush=True)
        for k, v in r.items():
            if k.startswith("init/") or k in ("rng_state_at_loop", "biases", "test_psnr", "seed"):
                out[f"s{s}/{k}"] = v
            else:
                out[k] = v                       # data set + configuration: identical for every seed
    out["n_seeds"] = np.asarray(a.first_seed + a.seeds)
    np.savez_compressed(a.out, **out)
    print(f"wrote {a.out} ({os.path.getsize(a.out) / 1e6:.2f} MB)")
    ps = np.stack([out[f"s{s}/test_psnr"] for s in range(a.first_seed + a.seeds)])          # [seed, eval, view]
    print("reference test PSNR, mean over views, per seed and evaluation:\n", np.round(ps.mean(-1), 3))


if __name__ == "__main__":
    main()
