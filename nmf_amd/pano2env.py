"""Fit an IntegralEquirect environment map to an equirectangular panorama -- counterpart of the reference's
scripts/pano2cube.py:1-146 (the tool that produces `backgrounds/forest.th` for `render_only=True fixed_bg=...`,
train.py:96-138):

    python -m nmf_amd.pano2env backgrounds/forest.exr --output log/forest.th [--res 1024] [--epochs 1000]

Same recipe: an `IntegralEquirect(bg_resolution=res, mipbias=0, activation='exp', lr=1e-3, init_val=-1.897, mul_lr=1e-3,
brightness_lr=0, betas=[0,0])`, Adam over its param groups with cosine annealing (eta_min 0.01), batches of 4096*50 random
panorama pixels looked up along their directions at a sharp footprint (log-solid-angle log 1e-5), Huber loss; the state_dict is
written with torch.save and the fitted map as `<prefix>pano.exr` next to it.  The panorama is read by nmf_amd.exr (the
image has no imageio / OpenEXR; NONE / RLE / ZIP(S) / DWAA / DWAB compressed files, i.e. the reference's backgrounds/*.exr as
they are).  `render.py --fixed-bg` loads the result at ITS OWN
resolution (the reference hard-codes 512 there while this tool's default is 1024)."""
import argparse
import json
import math
import os

import numpy as np
import torch

from . import exr
from .modules.integral_equirect import IntegralEquirect


def pixel_directions(rows, cols, H, W):
    """scripts/pano2cube.py:101-109: pixel (row, col) -> unit direction of the panorama parameterisation"""
    theta = rows / (H - 1) * math.pi - math.pi / 2
    phi = -cols / (W - 1) * 2 * math.pi - math.pi
    return torch.stack([torch.cos(phi) * torch.cos(theta), torch.sin(phi) * torch.cos(theta), -torch.sin(theta)], dim=1)


def fit(pano, res=1024, epochs=1000, batch_size=4096 * 50, device="cuda", seed=0, log=None):
    """pano [H,W,3] float (numpy / tensor) -> fitted IntegralEquirect, final PSNR-like figure of the script"""
    dev = torch.device(device)
    colors = torch.as_tensor(np.ascontiguousarray(pano), dtype=torch.float32, device=dev)
    H, W, _ = colors.shape
    colors = colors.reshape(-1, 3)
    N = colors.shape[0]
    bg = IntegralEquirect(bg_resolution=res, mipbias=0, activation="exp", lr=0.001, init_val=-1.897, mul_lr=0.001,
                          brightness_lr=0, betas=[0.0, 0.0], mul_betas=[0.9, 0.9], mipbias_lr=1e-4, mipnoise=0.0).to(dev)
    optim = torch.optim.Adam(bg.get_optparam_groups(), lr=0.001)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(optim, T_max=epochs, eta_min=0.01)
    loss_fn = torch.nn.HuberLoss()
    g = torch.Generator(device=dev).manual_seed(seed)
    batch = min(batch_size, N)
    ids, cur = None, N
    psnr = float("nan")
    sa = math.log(1e-5)
    for it in range(epochs):
        cur += batch                                              # SimpleSampler.nextids (:77-92)
        if ids is None or cur + batch > N:
            ids, cur = torch.randperm(N, device=dev, generator=g), 0
        inds = ids[cur:cur + batch]
        rows, cols = torch.div(inds, W, rounding_mode="floor").float(), (inds % W).float()
        vecs = pixel_directions(rows, cols, H, W)
        out = bg(vecs, torch.full((inds.shape[0],), sa, device=dev))
        samp = colors[inds]
        loss = loss_fn(out, samp)
        loss.backward()
        optim.step()
        optim.zero_grad()
        sched.step()
        if log is not None and (it % 100 == 0 or it == epochs - 1):
            photo = torch.sqrt((out.detach().clip(0, 1) - samp.clip(0, 1)) ** 2 + 1e-8).mean()
            psnr = -10.0 * math.log10(float(photo))
            log(dict(iteration=it, loss=float(loss.detach()), psnr=round(psnr, 3)))
    return bg, psnr


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("input")
    ap.add_argument("--output", default="log/mats360_bg.th")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--epochs", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=4096 * 50)
    args = ap.parse_args(argv)
    if args.input.lower().endswith(".exr"):
        pano = exr.imread(args.input)[..., :3]
    elif args.input.lower().endswith(".npy"):
        pano = np.load(args.input)[..., :3]
    else:
        from PIL import Image
        pano = np.asarray(Image.open(args.input).convert("RGB"), dtype=np.float32) / 255.0
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    bg, psnr = fit(pano, args.res, args.epochs, args.batch, dev, log=lambda r: print(json.dumps(r), flush=True))
    os.makedirs(os.path.dirname(os.path.abspath(args.output)), exist_ok=True)
    torch.save(bg.state_dict(), args.output)
    stem = os.path.splitext(os.path.basename(args.output))[0]
    bg.save(os.path.dirname(os.path.abspath(args.output)), prefix=stem + "_")
    rec = dict(output=args.output, resolution=args.res, panorama=list(pano.shape), psnr=psnr)
    print(json.dumps(rec), flush=True)
    return rec


if __name__ == "__main__":
    main()
