"""In-process A/B of the evaluation render: whole frames with an environment knob alternating frame by frame.
    python tools/ab_infer.py NAME v0,v1 [frames per value]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from nmf_amd import synthetic  # noqa: E402
from nmf_amd.noise import DeviceNoise  # noqa: E402
from nmf_amd.renderer import render_images  # noqa: E402

name, vals = sys.argv[1], sys.argv[2].split(",")
n = int(sys.argv[3]) if len(sys.argv) > 3 else 6
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
nerf, _ = bench.build(dev)
nerf.eval()
rays, focal = synthetic.camera_rays(0, all_pixels=True, wh=bench.FRAME)
rays = rays.to(dev)
noise = DeviceNoise(dev, seed=11)
chunk = nerf.eval_batch_size
for v in vals:
    os.environ[name] = v
    render_images(nerf, rays, focal, chunk, noise)
torch.cuda.synchronize()
t = {v: [] for v in vals}
for _ in range(n):
    for v in vals:
        os.environ[name] = v
        t0 = time.perf_counter()
        render_images(nerf, rays, focal, chunk, noise)
        torch.cuda.synchronize()
        t[v].append(time.perf_counter() - t0)
for v in vals:
    print(f"{name}={v}: median {statistics.median(t[v]) * 1e3:7.2f} ms / frame, {rays.shape[0] / statistics.median(t[v]) / 1e6:5.2f} M rays/s  "
          + " ".join(f"{x * 1e3:.1f}" for x in t[v]))
