"""Which (small) configuration of the S2 orbit data set trains to a given PSNR in a few hundred iterations?  Used to size the
reference run of tests/golden/make_psnr_trace.py (the reference trains on the CPU of the build container: minutes per seed).

    python tools/psnr_probe.py --grid 48 --bg 32 --res 32 --batch 1024 --rpr 32 --max-samples 40000 --iters 300
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nmf_amd import synthetic  # noqa: E402
from nmf_amd.config import build_model, resolved_config  # noqa: E402
from nmf_amd.noise import DeviceNoise  # noqa: E402
from nmf_amd.renderer import psnr_8bit, render_images  # noqa: E402
from nmf_amd.trainer import Trainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=48)
    ap.add_argument("--teacher-grid", type=int, default=None)
    ap.add_argument("--bg", type=int, default=32)
    ap.add_argument("--res", type=int, default=32)
    ap.add_argument("--views", type=int, default=24)
    ap.add_argument("--test-views", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--rpr", type=int, default=32)
    ap.add_argument("--max-samples", type=int, default=40000)
    ap.add_argument("--brdf-rays", type=int, nargs=2, default=[80000, 40000])
    ap.add_argument("--tns", type=int, default=80000)
    ap.add_argument("--retrace", type=int, default=200)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--every", type=int, default=50)
    ap.add_argument("--lr-iters", type=int, default=None, help="n_iters of the lr schedule (default: --iters)")
    ap.add_argument("--seed", type=int, default=20211200)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    tg = a.teacher_grid or a.grid
    teacher, _ = build_model(grid=tg, bg_resolution=a.bg, device=dev)
    teacher.load_state_dict(synthetic.state_dict_s1(grid=tg, bg_resolution=a.bg, seed=0), strict=False)
    teacher.eval()
    teacher.sampler.update(teacher.rf, init=False)
    teacher.sampler.update(teacher.rf, init=True)
    rays_tr, focal = synthetic.orbit_rays(a.views, a.res, seed=1)
    rays_te, _ = synthetic.orbit_rays(a.test_views, a.res, seed=2)
    rays_tr, rays_te = rays_tr.to(dev), rays_te.to(dev)
    gtn = DeviceNoise(dev, seed=7)
    rgb_tr = render_images(teacher, rays_tr, focal, 4096, gtn, draw_debug=True)
    rgb_te = render_images(teacher, rays_te, focal, 4096, gtn, draw_debug=True)
    del teacher
    torch.manual_seed(a.seed)
    over = {"sampler.max_samples": a.max_samples, "model.max_brdf_rays": list(a.brdf_rays), "model.target_num_samples": [a.tns],
            "model.max_retrace_rays": [a.retrace], "model.rays_per_ray": a.rpr, "sampler.update_list": [10 ** 9],
            "rf.upsamp_list": [10 ** 9]}
    nerf, _ = build_model(grid=a.grid, bg_resolution=a.bg, device=dev, overrides=over)
    nerf.train()
    params = dict(resolved_config()["params"], n_iters=a.lr_iters or a.iters, batch_size=a.batch, min_batch_size=a.batch,
                  max_batch_size=2 * a.batch, starting_batch_size=100, target_num_samples=a.max_samples)
    with torch.no_grad():
        xyz = torch.rand(100000, 4, device=dev) * 2 - 1
        xyz[:, 3] *= 0
        nerf.model.calibrate(None, xyz, nerf.rf.compute_appfeature(xyz), nerf.bg_module.mean_color().mean())
    tr = Trainer(nerf, params)
    noise = DeviceNoise(dev, seed=1000)
    g = torch.Generator(device=dev).manual_seed(a.seed)
    n_total = rays_tr.shape[0]
    perm, cur = torch.randperm(n_total, device=dev, generator=g), 0
    out = []
    for it in range(a.iters):
        nb = tr.lbatch_size()
        if cur + nb > n_total:
            perm, cur = torch.randperm(n_total, device=dev, generator=g), 0
        ids = perm[cur:cur + nb]
        cur += nb
        st = tr.step(rays_tr[ids], rgb_tr[ids], focal, noise=noise, global_rays=nb)
        if (it + 1) % a.every == 0:
            nerf.eval()
            pred = render_images(nerf, rays_te, focal, 4096, noise, draw_debug=True)
            nerf.train()
            pv, gv = pred.reshape(a.test_views, -1, 3), rgb_te.reshape(a.test_views, -1, 3)
            out.append((it + 1, round(float(torch.stack([psnr_8bit(pv[i], gv[i]) for i in range(a.test_views)]).mean()), 2),
                        round(st["psnr"], 2), nb))
    print(json.dumps(dict(args=vars(a), curve=out)))


if __name__ == "__main__":
    main()
