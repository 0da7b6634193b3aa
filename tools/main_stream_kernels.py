"""the kernels of the MAIN stream of a steady-state training step (the dependency chain of the step), by time: live HIP-event
timing of every launch on its launching stream (nmf_set_launch_probe), split by stream"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from nmf_amd import hip
from nmf_amd.noise import DeviceNoise
from nmf_amd.trainer import Trainer
dev = torch.device("cuda", 0)
nerf, params = bench.build(dev)
tr = Trainer(nerf, params)
batches, f = bench.make_batches(nerf, 30, 4096, 0, dev, distinct=12)
nz = DeviceNoise(dev, seed=5)
fx = hip.HOST_EXT
for i in range(40): tr.step(*batches[i % 12], f, noise=nz, update_controllers=False, fixed_chunk=4096)
torch.cuda.synchronize()
N = 30
fx.kernel_timing_begin("", True)
for i in range(N): tr.step(*batches[i % 12], f, noise=nz, update_controllers=False, fixed_chunk=4096)
t = fx.kernel_timing_end()
main = str(torch.cuda.current_stream().cuda_stream)
streams = sorted({k.split("@")[1] for k in t if "@" in k and not k.startswith("@")})
for sid in streams:
    rows = sorted(((k.split("@")[0], v) for k, v in t.items() if "@" in k and not k.startswith("@") and k.endswith("@" + sid)), key=lambda kv: -kv[1][0])
    tot = sum(v[0] for _, v in rows)
    print(f"stream {sid}{' (MAIN)' if sid == main else ''}: {1e3 * tot / N:.1f} us/step in {sum(v[1] for _, v in rows) / N:.1f} launches")
    for name, (ms, n) in rows[: 40 if sid == main else 6]:
        print(f"   {name:34s} {1e3 * ms / N:7.1f} us/step  {n / N:4.1f} x")
