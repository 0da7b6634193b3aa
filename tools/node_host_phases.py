"""host time of the phases of a training step that enters the C++ pass through TensorNeRF.forward + backward() (no device syncs):
where a loop that is not the Trainer's fused chunk() spends its host time"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from nmf_amd.functional import LossMix, SquaredError
from nmf_amd.noise import DeviceNoise
from nmf_amd.optim import FusedAdam

dev = torch.device("cuda", 0)
nerf, params = bench.build(dev)
batches, f = bench.make_batches(nerf, 60, 4096, 0, dev, distinct=12)
opt = FusedAdam(nerf.get_optparam_groups(), betas=tuple(params["betas"]), eps=params["eps"], weight_decay=params["weight_decay"])
nz = DeviceNoise(dev, seed=5)
bg = torch.ones(3, device=dev)
one = torch.ones((), device=dev)
acc = [0.0] * 5
N = 200
for i in range(N + 40):
    if i == 40:
        torch.cuda.synchronize(); acc = [0.0] * 5; t_all = time.perf_counter()
    rays, gt = batches[i % 12]
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    ims, st = nerf(rays, f, bg_col=bg, is_train=True, ndc_ray=False, noise=nz)
    t1 = time.perf_counter()
    loss = SquaredError.apply(ims["rgb_map"], gt[: ims["rgb_map"].shape[0]])
    l1 = nerf.rf.density_L1()
    total = LossMix.apply(1.0 / 4096, [1.0, params["L1_weight_initial"], params["ori_lambda"], 2 * params["pred_lambda"]], loss, l1,
                          st["ori_terms"], st["acc_terms"])
    t2 = time.perf_counter()
    total.backward(one)
    t3 = time.perf_counter()
    opt.step()
    t4 = time.perf_counter()
    for k, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
        acc[k] += v
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / N
print("wall ms/step", round(1e3 * wall, 4), "host phases us: forward %.0f loss %.0f backward %.0f optimizer+prefetch %.0f" %
      tuple(1e6 * a / N for a in acc[:4]))
