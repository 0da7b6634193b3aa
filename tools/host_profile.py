"""Host-side cost of one steady-state training step: cProfile over N steps (the GPU work is asynchronous, so cumulative
times are host time: Python glue, autograd dispatch, C-ABI wrappers, launches, the four size read-backs).
    python tools/host_profile.py [steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from nmf_amd.noise import DeviceNoise  # noqa: E402
from nmf_amd.trainer import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
nerf, params = bench.build(dev)
tr = Trainer(nerf, params)
noise = DeviceNoise(dev, seed=1)
batches, focal = bench.make_batches(nerf, 16, bench.CHUNK, 0, dev, distinct=16)
for i in range(15):
    tr.step(*batches[i % 16], focal, noise=noise, update_controllers=False, fixed_chunk=bench.CHUNK)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    tr.step(*batches[i % 16], focal, noise=noise, update_controllers=False, fixed_chunk=bench.CHUNK)
torch.cuda.synchronize()
print(f"unprofiled: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step, load {os.getloadavg()}")
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    tr.step(*batches[i % 16], focal, noise=noise, update_controllers=False, fixed_chunk=bench.CHUNK)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
print(s.getvalue()[:9000])
s = io.StringIO()
st = pstats.Stats(pr, stream=s)
st.sort_stats("cumulative")
for fn in ("_core_tables", "end_step", "_step_planned", "step", "_core_chunk", "_core_sync"):
    st.print_callees(fn)
print(s.getvalue()[:14000])
