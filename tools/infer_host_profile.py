"""Host-side cost of the evaluation render (one 800 x 800 frame, eval_batch_size rays per chunk): cProfile over one frame;
the GPU work is asynchronous, so the times are host time (Python glue, noise callbacks, C-ABI wrappers, read-back waits).
    python tools/infer_host_profile.py"""
import cProfile
import io
import os
import pstats
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
chunk = nerf.eval_batch_size
render_images(nerf, rays[: 8 * chunk], focal, chunk, noise)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    render_images(nerf, rays, focal, chunk, noise)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"unprofiled: {dt * 1e3:.1f} ms / frame, {rays.shape[0] / dt / 1e6:.2f} M rays/s, {dt / (rays.shape[0] / chunk) * 1e6:.0f} us / chunk")
pr = cProfile.Profile()
pr.enable()
render_images(nerf, rays, focal, chunk, noise)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:7000])
