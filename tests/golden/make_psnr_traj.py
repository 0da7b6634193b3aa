"""Container-only: further seeds of the reference's 300-iteration S2 training (the configuration of make_psnr_trace.py /
make_psnr_more.py) that keep the run's TRAJECTORY next to the test PSNR: per chunk the loss the loop back-propagates, the ray
controller (`num_rays`), the re-trace controller (`max_retrace_rays`), the sample counts, per iteration the global batch, every
group's learning rate, every parameter's gradient norm and norm ("something between iterations 40 and 200
differs, or the reference's low tail explains it" -- 300 points per run instead of 3).  tests/golden/psnr_ref_traj_<tag>.npz holds
arrays only; tools/psnr_trajectory.py compares the seed means with the build's own runs.

    python tests/golden/make_psnr_traj.py --first-seed 200 --seeds 8 --stride 3 --threads 2 --out tests/golden/psnr_ref_traj_a.npz
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
    ap.add_argument("--first-seed", type=int, default=200)
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(HERE, "psnr_ref_traj.npz"))
    ap.add_argument("--override", action="append", default=[], help="extra hydra override of the reference run, e.g. model.params.ori_lambda=0 "
                    "(a factor experiment: give --out another name, the files psnr_ref_traj_*.npz are the unmodified configuration)")
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
                   max_brdf_rays=(80000, 40000), target_num_samples=80000, max_retrace=1000, rays_per_ray=128, light="traj",
                   threads=a.threads, extra_overrides=tuple(a.override))
        for k, v in r.items():
            if k in ("psnr_at", "gradnorm_names", "param_names"):
                done[k] = np.asarray(v)
            else:
                done[f"s{s}/{k}"] = np.asarray(v)
        np.savez_compressed(a.out, **done)
        print(f"seed {s}: {time.time() - t0:.0f} s, mean test PSNR {np.asarray(r['test_psnr']).mean(-1)}", flush=True)


if __name__ == "__main__":
    main()
