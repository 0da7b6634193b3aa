import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch, time, cProfile, pstats
from nmf_amd.noise import DeviceNoise
from nmf_amd.trainer import Trainer
from nmf_amd.optim import FusedAdam
dev=torch.device("cuda",0)
nerf, params = bench.build(dev)
def sync(): torch.cuda.synchronize()
batches, f = bench.make_batches(nerf, 60, 4096, 0, dev, distinct=12)
for tf in (True, False):
    tr = Trainer(nerf, params, tape_free=tf)
    dt, rays, last, _ = bench.time_train(tr, batches, f, DeviceNoise(dev, seed=5), 40, 100, 4096, sync)
    print("trainer tape_free" if tf else "trainer node", round(1e3*dt/100,4), flush=True)
opt = FusedAdam(nerf.get_optparam_groups(), betas=tuple(params["betas"]), eps=params["eps"], weight_decay=params["weight_decay"])
nz = DeviceNoise(dev, seed=5)
for i in range(20): bench.reference_style_step(nerf, opt, *batches[i%12], f, params, nz, 4096)
sync(); t0=time.perf_counter()
for i in range(60): bench.reference_style_step(nerf, opt, *batches[i%12], f, params, nz, 4096)
sync(); print("reference loop fused", 1e3*(time.perf_counter()-t0)/60, flush=True)
pr = cProfile.Profile(); pr.enable()
for i in range(40): bench.reference_style_step(nerf, opt, *batches[i%12], f, params, nz, 4096)
sync(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
