"""host time of the pieces of Trainer.step with the all-reduce entered on one rank (no device syncs inside the loop)"""
import os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NMF_ALLREDUCE_SINGLE_RANK"] = "1"
import torch
import torch.distributed as dist
import bench
from nmf_amd.noise import DeviceNoise
from nmf_amd.trainer import Trainer
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
dist.all_reduce(torch.zeros(1, device=dev))
nerf, params = bench.build(dev)
tr = Trainer(nerf, params)
batches, f = bench.make_batches(nerf, 60, 4096, 0, dev, distinct=12)
nz = DeviceNoise(dev, seed=5)
T = {}
def wrap(obj, name, key=None):
    fn = getattr(obj, name); key = key or name
    def w(*a, **k):
        t0 = time.perf_counter()
        try: return fn(*a, **k)
        finally: T[key] = T.get(key, 0.0) + time.perf_counter() - t0
    setattr(obj, name, w)
for i in range(40): tr.step(*batches[i % 12], f, noise=nz, update_controllers=False, fixed_chunk=4096)
wrap(tr.fast, "chunk"); wrap(tr.fast, "end_step"); wrap(tr.fast, "prefetch"); wrap(tr.fast, "early_pairs")
red = tr.reduce
wrap(red, "early_inplace"); wrap(red, "late_inplace"); wrap(red, "finish_early"); wrap(tr.optimizer, "step_unhooked"); wrap(dist, "all_reduce", "dist.all_reduce")
for mode in (True, False):
    red.single_rank = mode
    for i in range(20): tr.step(*batches[i % 12], f, noise=nz, update_controllers=False, fixed_chunk=4096)
    torch.cuda.synchronize(); T.clear(); N = 200; t0 = time.perf_counter()
    for i in range(N): tr.step(*batches[i % 12], f, noise=nz, update_controllers=False, fixed_chunk=4096)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print("all-reduce" if mode else "no collective", "wall ms/step %.4f, host loop %.4f" % (1e3 * (time.perf_counter() - t0) / N, 1e3 * (t1 - t0) / N))
    for k, v in sorted(T.items(), key=lambda kv: -kv[1]): print(f"   {k:20s} {1e6 * v / N:8.1f} us/step")
dist.destroy_process_group()
