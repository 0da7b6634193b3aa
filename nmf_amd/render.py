"""Inference entry point -- counterpart of the reference's `render_only=True` path (train.py:64-190 `render_test`,
renderer.py:56-106 `chunk_renderer`, :399-401 PSNR) for checkpoints written by TensorNeRF.save:

    python -m nmf_amd.render --ckpt log/lego.th --datadir /data/nerf_synthetic/lego [--fixed-bg forest.th] [--out imgs/]
    python -m nmf_amd.render --ckpt log/s1.th --views 4 --res 800                      (synthetic orbit cameras)

--fixed-bg swaps the learned environment map for another IntegralEquirect state_dict (relighting, train.py:96-138); the
module is rebuilt at the resolution stored in that file (the reference hard-codes 512 and fails on other sizes).
Prints one JSON line: frames, rays/s (render to completion, eval_batch_size rays per chunk), mean PSNR when ground truth exists.
"""
import argparse
import json
import os
import time

import torch

from . import synthetic
from .modules.integral_equirect import IntegralEquirect
from .modules.tensor_nerf import TensorNeRF
from .noise import DeviceNoise
from .renderer import psnr_8bit, render_images


def load_fixed_bg(path, device):
    """train.py:96-138: an IntegralEquirect with mipbias 0 / activation exp, its learning rates zeroed"""
    from .checkpoint import load_checkpoint
    sd = load_checkpoint(path)
    sd = sd.get("state_dict", sd) if isinstance(sd, dict) and "bg_mat" not in sd else sd
    res = int(sd["bg_mat"].shape[-2])
    bg = IntegralEquirect(bg_resolution=res, mipbias=0, activation="exp", lr=0.0, init_val=-1.897, mul_lr=0.0,
                          brightness_lr=0, betas=[0.0, 0.0], mul_betas=[0.9, 0.9], mipbias_lr=0.0, mipnoise=0.0)
    bg.load_state_dict({k: v for k, v in sd.items() if k in bg.state_dict()}, strict=False)
    return bg.to(device)


@torch.no_grad()
def render_frames(nerf, rays, focal, chunk, noise):
    """rays [F, h*w, 6] -> rgb [F, h*w, 3], seconds; render2completion over `chunk`-ray slices (renderer.py:56-106)"""
    out = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(rays.shape[0]):
        out.append(render_images(nerf, rays[f], focal, chunk, noise))
    torch.cuda.synchronize()
    return torch.stack(out), time.perf_counter() - t0


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", required=True)
    ap.add_argument("--datadir", default=None)
    ap.add_argument("--near-far", type=float, nargs=2, default=[2.5, 7.0])
    ap.add_argument("--fixed-bg", default=None)
    ap.add_argument("--out", default=None, help="directory for PNG frames")
    ap.add_argument("--views", type=int, default=2)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--n-vis", type=int, default=-1)
    ap.add_argument("--chunk", type=int, default=None, help="rays per chunk (default: the model's eval_batch_size)")
    args = ap.parse_args(argv)
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    gt = None
    if args.datadir:
        from .dataLoader import BlenderDataset
        ds = BlenderDataset(args.datadir, split="test", is_stack=True, N_vis=args.n_vis)
        rays, focal, near_far = ds.all_rays.to(dev), float(ds.fx), tuple(ds.near_far)
        gt = ds.all_rgbs.reshape(rays.shape[0], -1, 3).to(dev)
        wh = ds.img_wh
    else:
        r, focal = synthetic.orbit_rays(args.views, args.res, seed=2)
        rays, near_far, wh = r.reshape(args.views, -1, 6).to(dev), tuple(args.near_far), [args.res, args.res]
    nerf = TensorNeRF.load(args.ckpt, near_far=list(near_far), device=dev)
    if args.fixed_bg:
        nerf.bg_module = load_fixed_bg(args.fixed_bg, dev)
    nerf.eval()
    chunk = args.chunk or nerf.eval_batch_size
    noise = DeviceNoise(dev, seed=11)
    render_frames(nerf, rays[:1, : min(chunk, rays.shape[1])], focal, chunk, noise)          # warm-up (table builds)
    rgb, dt = render_frames(nerf, rays, focal, chunk, noise)
    rec = dict(frames=int(rays.shape[0]), width=wh[0], height=wh[1], chunk=chunk, seconds=round(dt, 4),
               rays_per_s=round(rays.shape[0] * rays.shape[1] / dt, 1), relit=bool(args.fixed_bg))
    if gt is not None:
        rec["psnr"] = round(float(torch.stack([psnr_8bit(rgb[i], gt[i]) for i in range(rgb.shape[0])]).mean()), 3)
    if args.out:
        from PIL import Image
        os.makedirs(args.out, exist_ok=True)
        for i in range(rgb.shape[0]):
            a = (rgb[i].clip(0, 1).reshape(wh[1], wh[0], 3) * 255).byte().cpu().numpy()
            Image.fromarray(a).save(os.path.join(args.out, f"{i:03d}.png"))
    print(json.dumps(rec), flush=True)
    return rec


if __name__ == "__main__":
    main()
