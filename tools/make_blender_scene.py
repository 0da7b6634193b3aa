"""Writes a scene directory in the nerf_synthetic / Blender format -- transforms_{train,test}.json + RGBA PNG frames -- rendered
from the synthetic scene S1, so that the REAL-data path (dataLoader/blender.py -> `python -m nmf_amd.train dataset=...`)
runs at the size of the reference's data sets without them being on the box: nerf_synthetic itself is not available offline.

    python tools/make_blender_scene.py --out /tmp/ns/lego --views 100 --test-views 8 --res 800
    python -m nmf_amd.train dataset=lego datadir=/tmp/ns expname=s1_as_lego --iters 300 --eval-every 300

Cameras sit on the upper hemisphere of radius 4 looking at the origin (camera_angle_x as in the lego set); a frame stores the
straight (un-premultiplied) colour and the accumulated opacity as alpha, which is what the loader blends onto white
(train.py:525-530).  The rays a frame is rendered with are the ones BlenderDataset builds from the pose that is written."""
import argparse
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def look_at_blender(eye):
    """camera-to-world of a Blender camera (looks down -z, y up) at `eye` looking at the origin"""
    eye = np.asarray(eye, dtype=np.float64)
    fwd = -eye / np.linalg.norm(eye)
    right = np.cross(fwd, [0.0, 0.0, 1.0])
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up, -fwd, eye
    return c2w


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--views", type=int, default=100)
    ap.add_argument("--test-views", type=int, default=8)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--bg", type=int, default=512)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)
    from PIL import Image
    from nmf_amd import synthetic
    from nmf_amd.config import build_model
    from nmf_amd.dataLoader.blender import BLENDER2OPENCV, get_ray_directions, get_rays
    from nmf_amd.noise import DeviceNoise
    from nmf_amd.renderer import render_images
    dev = torch.device("cuda", 0)
    nerf, _ = build_model(grid=args.grid, bg_resolution=args.bg, device=dev)
    nerf.load_state_dict(synthetic.state_dict_s1(grid=args.grid, bg_resolution=args.bg, seed=0), strict=False)
    nerf.eval()
    nerf.sampler.update(nerf.rf, init=False)
    nerf.sampler.update(nerf.rf, init=True)
    angle = 0.6911112070083618                      # camera_angle_x of nerf_synthetic/lego
    w = h = args.res
    fx = 0.5 * w / math.tan(0.5 * angle)
    dirs = get_ray_directions(h, w, [fx, fx])
    dirs = dirs / torch.norm(dirs, dim=-1, keepdim=True)
    rng = np.random.default_rng(args.seed)
    noise = DeviceNoise(dev, seed=3)
    for split, n in (("train", args.views), ("test", args.test_views)):
        os.makedirs(os.path.join(args.out, split), exist_ok=True)
        frames = []
        for i in range(n):
            z = rng.uniform(0.15, 0.9)
            phi = rng.uniform(0, 2 * math.pi)
            rxy = math.sqrt(1 - z * z)
            c2w = look_at_blender(4.0 * np.array([rxy * math.cos(phi), rxy * math.sin(phi), z]))
            o, d = get_rays(dirs, torch.FloatTensor(c2w @ BLENDER2OPENCV))
            rays = torch.cat([o, d], 1).to(dev)
            out = render_images(nerf, rays, fx, 32768, noise, keys=("rgb_map", "acc_map"))
            acc = out["acc_map"].clamp(0, 1).reshape(h, w, 1)
            comp = out["rgb_map"].reshape(h, w, 3)                       # colour over the white background
            straight = ((comp - (1 - acc)) / acc.clamp_min(1e-4)).clamp(0, 1) * (acc > 1e-4)
            rgba = torch.cat([straight, acc], -1).mul(255).round().byte().cpu().numpy()
            Image.fromarray(rgba, "RGBA").save(os.path.join(args.out, split, f"r_{i}.png"))
            frames.append({"file_path": f"./{split}/r_{i}", "transform_matrix": c2w.tolist()})
        json.dump({"camera_angle_x": angle, "w": w, "h": h, "frames": frames}, open(os.path.join(args.out, f"transforms_{split}.json"), "w"))
    print(f"wrote {args.views} + {args.test_views} frames of {w} x {h} to {args.out}")


if __name__ == "__main__":
    main()
