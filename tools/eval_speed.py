"""Inference rays/s of whole 800 x 800 frames against `eval_batch_size` (the reference's config key, 4096 in
configs/model/microfacet_tensorf2.yaml): one warm-up frame per chunk size (allocator growth), two timed.
    python tools/eval_speed.py [chunk ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from nmf_amd import synthetic  # noqa: E402
from nmf_amd.noise import DeviceNoise  # noqa: E402
from nmf_amd.renderer import render_images  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
nerf, _ = bench.build(dev)
nerf.eval()
rays, focal = synthetic.camera_rays(0, all_pixels=True, wh=bench.FRAME)
rays = rays.to(dev)
noise = DeviceNoise(dev, seed=11)
for chunk in [int(a) for a in sys.argv[1:]] or (4096, 8192, 16384, 32768):
    ref = render_images(nerf, rays, focal, chunk, noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        rgb = render_images(nerf, rays, focal, chunk, noise)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    print(f"eval_batch_size {chunk:6d}: {rays.shape[0] / dt / 1e6:6.2f} M rays/s, {dt * 1e3:7.1f} ms / frame, "
          f"mean |rgb - previous frame| {float((rgb - ref).abs().mean()):.2e}")
