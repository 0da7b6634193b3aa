"""Container-only: MORE seeds of the reference's 300-iteration S2 training (the configuration of make_psnr_trace.py / psnr_trace.npz),
of which only the test PSNR per view is kept (tests/golden/psnr_ref_more.npz, a few KB): the comparison of the PSNR after equal
iterations is a comparison of two distributions over seeds, and its resolution is set by the number of REFERENCE runs (11-17 minutes
each on 4 cores).  Written after every seed, so that a run can be stopped at any time.

    python tests/golden/make_psnr_more.py --first-seed 6 --seeds 20 --stride 2 [--out tests/golden/psnr_ref_more_a.npz]
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
    ap.add_argument("--first-seed", type=int, default=6)
    ap.add_argument("--seeds", type=int, default=20)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--out", default=os.path.join(HERE, "psnr_ref_more.npz"))
    a = ap.parse_args()
    done = {}
    if os.path.exists(a.out):
        with np.load(a.out) as old:
            done = {k: old[k] for k in old.files}
    for i in range(a.seeds):
        s = a.first_seed + i * a.stride
        if f"s{s}/test_psnr" in done:
            continue
        t0 = time.time()
        r = mt.run(grid0=48, grid1=48, teacher_grid=48, bg=32, upsample_at=(1000000,), n_iters=30000, stop_at=300, psnr_at=(100, 200, 300),
                   res=32, train_views=24, test_views=3, seed=20211200 + s, batch=1024, max_batch=2048, max_samples=40000,
                   max_brdf_rays=(80000, 40000), target_num_samples=80000, max_retrace=1000, rays_per_ray=128, light=True,
                   threads=a.threads)
        done[f"s{s}/test_psnr"] = np.asarray(r["test_psnr"])
        done["psnr_at"] = np.asarray(r["psnr_at"]) if "psnr_at" in r else np.asarray([100, 200, 300])
        np.savez_compressed(a.out, **done)
        print(f"seed {s}: {time.time() - t0:.0f} s, mean test PSNR {np.asarray(r['test_psnr']).mean(-1)}", flush=True)


if __name__ == "__main__":
    main()
