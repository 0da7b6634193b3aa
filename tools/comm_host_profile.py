"""cProfile of Trainer.step with the gradient all-reduce entered on a process group of ONE rank (RCCL): what the data-parallel path
costs the host per step next to the single-process step"""
import cProfile, os, pstats, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NMF_ALLREDUCE_SINGLE_RANK"] = "1"
import torch
import torch.distributed as dist
import bench
from nmf_amd.noise import DeviceNoise
from nmf_amd.trainer import Trainer

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
dist.all_reduce(torch.zeros(1, device=dev))
nerf, params = bench.build(dev)
tr = Trainer(nerf, params)
batches, f = bench.make_batches(nerf, 60, 4096, 0, dev, distinct=12)
nz = DeviceNoise(dev, seed=5)
def run(n):
    for i in range(n):
        tr.step(*batches[i % 12], f, noise=nz, update_controllers=False, fixed_chunk=4096)
run(40); torch.cuda.synchronize()
t0 = time.perf_counter(); run(100); torch.cuda.synchronize(); print("ms/step with the all-reduce", 1e3 * (time.perf_counter() - t0) / 100)
tr.reduce.single_rank = False
run(20); torch.cuda.synchronize()
t0 = time.perf_counter(); run(100); torch.cuda.synchronize(); print("ms/step without", 1e3 * (time.perf_counter() - t0) / 100)
tr.reduce.single_rank = True
run(10)
pr = cProfile.Profile(); pr.enable(); run(60); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
dist.destroy_process_group()
