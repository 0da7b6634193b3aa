"""How long a training step takes when the device has (almost) nothing to do: the same launch sequence on tiny chunks -- the host's
issue time + the latency of the dependent launches and read-backs, i.e. the floor no kernel optimisation can get under.
    python tools/host_floor.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from nmf_amd.noise import DeviceNoise  # noqa: E402
from nmf_amd.trainer import Trainer  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
nerf, params = bench.build(dev)
tr = Trainer(nerf, params)
noise = DeviceNoise(dev, seed=1)
for rays_n in (64, 512, 4096):
    batches, focal = bench.make_batches(nerf, 8, rays_n, 0, dev, distinct=8)
    for i in range(30):
        tr.step(*batches[i % 8], focal, noise=noise, update_controllers=False, fixed_chunk=rays_n)
    torch.cuda.synchronize()
    from nmf_amd import hip
    hip.HOST_EXT.readback_wait_us(True)
    hip.HOST_EXT.readback_wait_by_slot_us(True)
    t0, c0 = time.perf_counter(), time.process_time()
    n = 100
    for i in range(n):
        tr.step(*batches[i % 8], focal, noise=noise, update_controllers=False, fixed_chunk=rays_n)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    wait = hip.HOST_EXT.readback_wait_us(True) / n
    print("      waits by read-back (level-0 samples, level-0 rows, level-1 samples, level-1 rows), us:",
          [round(w / n, 1) for w in hip.HOST_EXT.readback_wait_by_slot_us(True)])
    print(f"{rays_n:5d} rays per step: {dt / n * 1e3:.3f} ms per step (issue loop alone {t_issue / n * 1e3:.3f} ms of which {wait / 1e3:.3f} ms "
          f"waiting for the four size read-backs -> host busy {(t_issue / n * 1e3 - wait / 1e3):.3f} ms; process CPU "
          f"{(time.process_time() - c0) / n * 1e3:.3f} ms), sizes {tr.fast.last_sizes}")
