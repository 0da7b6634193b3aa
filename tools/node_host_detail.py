import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from nmf_amd import fast_step
from nmf_amd.functional import LossMix, SquaredError
from nmf_amd.noise import DeviceNoise
from nmf_amd.optim import FusedAdam
T = {}
def wrap(obj, name, key=None):
    fn = getattr(obj, name); key = key or name
    def w(*a, **k):
        t0 = time.perf_counter()
        try: return fn(*a, **k)
        finally: T[key] = T.get(key, 0.0) + time.perf_counter() - t0
    setattr(obj, name, w)
dev = torch.device("cuda", 0)
nerf, params = bench.build(dev)
batches, f = bench.make_batches(nerf, 60, 4096, 0, dev, distinct=12)
opt = FusedAdam(nerf.get_optparam_groups(), betas=tuple(params["betas"]), eps=params["eps"], weight_decay=params["weight_decay"])
nz = DeviceNoise(dev, seed=5); bg = torch.ones(3, device=dev); one = torch.ones((), device=dev)
def step(i):
    rays, gt = batches[i % 12]
    opt.zero_grad(set_to_none=True)
    ims, st = nerf(rays, f, bg_col=bg, is_train=True, ndc_ray=False, noise=nz)
    loss = SquaredError.apply(ims["rgb_map"], gt[: ims["rgb_map"].shape[0]])
    total = LossMix.apply(1.0 / 4096, [1.0, params["L1_weight_initial"], params["ori_lambda"], 2 * params["pred_lambda"]], loss, nerf.rf.density_L1(), st["ori_terms"], st["acc_terms"])
    total.backward(one)
    opt.step()
for i in range(30): step(i)
tp = nerf._fused_pass
wrap(tp, "backward_autograd"); wrap(tp, "_finish_grads"); wrap(tp, "prefetch"); wrap(tp, "_core_sync"); wrap(tp, "forward_autograd")
c = tp.core()
class CW:
    def __init__(s, c): object.__setattr__(s, "c", c)
    def __getattr__(s, n):
        v = getattr(s.c, n)
        if n in ("train_backward", "train_forward"):
            def w(*a, **k):
                t0 = time.perf_counter()
                try: return v(*a, **k)
                finally: T[n] = T.get(n, 0.0) + time.perf_counter() - t0
            return w
        return v
    def __setattr__(s, n, v): setattr(s.c, n, v)
tp._core = tp.context(0).core = CW(c)          # (round 6: the pass reaches its StepCore through its chunk context 0)
wrap(opt, "_step_planned")
from nmf_amd import hip
for n in ("vm_pack_density", "brdf_mlp_pack", "sat_build", "sh_project", "sat_lookup_fwd", "vm_unpack_density_grad", "multi_copy"):
    if hasattr(hip, n): wrap(hip, n, "hip." + n)
wrap(nerf.model.diffuse_module, "head_pass"); wrap(nerf.model.brdf, "mlp_pass"); wrap(nerf.bg_module, "_tables", "bg._tables"); wrap(nerf.bg_module, "get_spherical_harmonics"); wrap(nerf.rf, "_fwd_tables")
torch.cuda.synchronize(); N = 200; t0 = time.perf_counter()
for i in range(N): step(i)
torch.cuda.synchronize(); print("wall", 1e3 * (time.perf_counter() - t0) / N)
for k, v in sorted(T.items(), key=lambda kv: -kv[1]): print(f"{k:32s} {1e6 * v / N:8.1f} us/step")
