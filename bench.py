#!/usr/bin/env python
"""Benchmark of the microfacet_tensorf2 hot path on MI355X (see BASELINE.json / SURVEY.md 8d).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one optimizer step (forward + backward + Adam) over a 4096-ray batch per GPU of the synthetic
scene S1 (solid cube in the 128^3 TensoRF grid, 512x1024 env map, 800x800 camera; nerf_synthetic/lego is not
available offline) in the STEADY-STATE phase of the reference (every secondary ray re-traced).
Ray batches are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RAYS_PER_GPU = 4096
GRID = 128
BG_RES = 512
# Algorithmic bytes per kept sample, forward (SURVEY 8d, fp32 tables, 18 taps): density value 1152 + density gradient
# 1920 + appearance 1728 = G_s = 4800.  Backward = recompute read + read-modify-write of the gradients = 3 x forward.
# The density / normal walk runs over all kept samples, the appearance walk only over the bounce rows (sparse appearance).
G_DENSITY, G_APP = 1152 + 1920, 1728
BWD_BYTES_DENSITY, BWD_BYTES_APP = 3 * G_DENSITY, 3 * G_APP
HBM_PEAK_GBS = 8000.0
# the same kernel seen from the matrix pipe: per 4 samples 3 x 20 (density) / 3 x 18 (appearance) v_mfma_f32_16x16x4_f32 of
# 2048 FLOP each (dense-equivalent, ~95 % of the products are structural zeros of the scatter matrix)
MFMA_FLOP_DENSITY, MFMA_FLOP_APP = 3 * 20 * 2048 / 4, 3 * 18 * 2048 / 4
MFMA_F32_PEAK_TFLOPS = 157.3


def build(device):
    from nmf_amd import synthetic
    from nmf_amd.config import build_model, resolved_config
    nerf, cfg = build_model(grid=GRID, bg_resolution=BG_RES, device=device)
    nerf.load_state_dict(synthetic.state_dict_s1(grid=GRID, bg_resolution=BG_RES, seed=0), strict=False)
    nerf.train()
    nerf.sampler.update(nerf.rf, init=False)          # alpha mask from the density field (alphagrid.py:250-276)
    nerf.sampler.update(nerf.rf, init=True)
    nerf.model.detach_N = False                        # state after the first check_schedule (microfacet.py:117)
    nerf.model.max_retrace_rays = [nerf.model.max_brdf_rays[0]]   # steady state: all secondary rays re-traced
    return nerf, resolved_config()["params"]


class KernelTimer:
    """HIP events around every nmf_vm_query_bwd / nmf_vm_query_bwd_segments call (issued on torch's current stream, which is the stream the
    C ABI launches on) -> per-launch duration of the dominant kernel inside the timed region."""

    def __init__(self):
        from nmf_amd import hip, functional
        self.hip = hip
        self.records = []
        self.enabled = False
        orig = hip.vm_query_bwd

        def wrapped(p, xyzt, *a, **k):
            if not self.enabled:
                return orig(p, xyzt, *a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(p, xyzt, *a, **k)
            e.record()
            dens = a[7] is not None or a[8] is not None or a[9] is not None      # d_sigma / d_sigma_feat / d_normal
            app = a[10] is not None                                              # d_app
            self.records.append((s, e, int(xyzt.shape[0]), dens, app))
            return r

        hip.vm_query_bwd = wrapped
        functional.hip.vm_query_bwd = wrapped
        orig_segs = hip.vm_query_bwd_segments

        def wrapped_segs(p, segs, *a, **k):      # the training pass walks its sample sets together (functional.py)
            if not self.enabled:
                return orig_segs(p, segs, *a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_segs(p, segs, *a, **k)
            e.record()
            m = sum(int(sg[0].shape[0]) for sg in segs)
            dens = any(sg[3] is not None or sg[4] is not None or sg[5] is not None for sg in segs)
            app = any(sg[6] is not None for sg in segs)
            self.records.append((s, e, m, dens, app))
            return r

        hip.vm_query_bwd_segments = wrapped_segs
        # per-step derived tables must be rebuilt after every optimizer update (packed density planes, SAT): count the
        # rebuilds inside the timed region so a stale cache (= skipped work) shows up in the report
        self.rebuilds = {"vm_pack_density": 0, "sat_build": 0}
        for name in self.rebuilds:
            fn = getattr(hip, name)

            def counted(*a, _fn=fn, _name=name, **k):
                if self.enabled:
                    self.rebuilds[_name] += 1
                return _fn(*a, **k)

            setattr(hip, name, counted)

    def summary(self):
        """-> total ms, algorithmic bytes, dense-equivalent MFMA flop, samples, launches"""
        ms = sum(r[0].elapsed_time(r[1]) for r in self.records)
        nbytes = sum(m * (BWD_BYTES_DENSITY * d + BWD_BYTES_APP * a) for _, _, m, d, a in self.records)
        flop = sum(m * (MFMA_FLOP_DENSITY * d + MFMA_FLOP_APP * a) for _, _, m, d, a in self.records)
        return ms, nbytes, flop, sum(r[2] for r in self.records), len(self.records)


def cpu_baseline(n_rays=512):
    """The CPU oracle (validated against the reference, tests/test_oracle_golden.py) timed on the host cores:
    forward + backward of one steady-state chunk on a bounded sample of the same workload."""
    from nmf_amd import synthetic
    from oracle import nmf_oracle as O
    torch.manual_seed(0)
    sd = synthetic.state_dict_s1(grid=GRID, bg_resolution=BG_RES, seed=0)
    for k, v in sd.items():
        if k != "model.brdf_sampler.angs":
            v.requires_grad_(True)
    cfg = O.Cfg(grid=GRID, detach_N=False, max_retrace_rays=(650000,))
    vol = O.dense_alpha_mask({k: v.detach() for k, v in sd.items()}, cfg)
    rays, focal = synthetic.camera_rays(n_rays, seed=123)
    gt = torch.rand(n_rays, 3)
    t0 = time.time()
    ims, st = O.render(sd, cfg, rays, focal, vol, O.Noise(), is_train=True, bg_col=torch.ones(3))
    total, _ = O.training_loss(ims, st, gt, n_rays, sd)
    total.backward()
    dt = time.time() - t0
    return dict(value=n_rays / dt, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                sample=f"1 steady-state chunk of {n_rays} rays (S1, 128^3, all {st['n_samples']} samples incl. re-traced "
                       f"secondary rays), forward+backward, {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with torch.distributed.run (see the module docstring)")
    # NMF_BENCH_SHARE_GPU=1 + NMF_BENCH_BACKEND=gloo: functional test of the multi-process path on a 1-GPU box
    # (RCCL refuses two ranks on one device); the driver's real runs use one GPU per rank over RCCL/xGMI.
    if os.environ.get("NMF_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("NMF_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)
        # create the communicator (RCCL ring / tree setup takes seconds) outside every timed or warm-up step
        dist.all_reduce(torch.zeros(1, device=device))
        torch.cuda.synchronize()

    import __graft_entry__ as ge
    if rank == 0 and not os.path.exists(os.path.join(ROOT, "nmf_amd", "lib", "libnmf_hip.so")):
        ge.build()
    if world > 1:
        dist.barrier()
    from nmf_amd import synthetic
    from nmf_amd.noise import DeviceNoise
    from nmf_amd.trainer import Trainer

    torch.manual_seed(20211200)
    nerf, params = build(device)
    trainer = Trainer(nerf, params, world_size=world, rank=rank)
    noise = DeviceNoise(device, seed=1000 + rank)
    timer = KernelTimer()

    n_steps = args.warmup + args.steps
    batches = []
    g = torch.Generator().manual_seed(77 + rank)
    for i in range(n_steps):                       # disjoint random pixels per rank and step, resident in HBM
        rays, focal = synthetic.camera_rays(RAYS_PER_GPU, seed=10007 * (rank + 1) + i)
        batches.append((rays.to(device), torch.rand(RAYS_PER_GPU, 3, generator=g).to(device)))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        trainer.step(*batches[i], focal, noise=noise, update_controllers=False, fixed_chunk=RAYS_PER_GPU)
    sync()
    timer.enabled = True
    t0 = time.perf_counter()
    rays_done, last = 0, None
    for i in range(args.warmup, n_steps):
        last = trainer.step(*batches[i], focal, noise=noise, update_controllers=False, fixed_chunk=RAYS_PER_GPU)
        rays_done += last["rays"]
    sync()
    dt = time.perf_counter() - t0
    timer.enabled = False

    tt = torch.tensor([dt, float(rays_done)], dtype=torch.float64, device=device)
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt_max, rays_all = float(tmax[0]), float(tt[1])
    else:
        dt_max, rays_all = dt, float(rays_done)

    if min(timer.rebuilds.values()) < args.steps:
        raise SystemExit(f"derived tables were not rebuilt every step: {timer.rebuilds} for {args.steps} steps")
    if rank == 0:
        k_ms, k_bytes, k_flop, k_samples, k_launches = timer.summary()
        achieved = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_vm_bwd.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        out = {
            "metric": "train rays/sec (microfacet_tensorf2, 4096-ray batch per GPU, steady state)",
            "value": rays_all / dt_max, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "S1 solid-cube scene, TensoRF 128^3 (16+24 comps), env 512x1024, 800x800 camera, "
                                   "4096 rays/GPU/step, fwd+bwd+Adam, all secondary rays re-traced (steady state); "
                                   "stands in for BASELINE configs[1] (lego is not available offline)",
                       "rays_per_gpu": RAYS_PER_GPU, "grid": GRID, "samples_per_step": last["n_samples"],
                       "table_rebuilds_in_timed_region": dict(timer.rebuilds),
                       "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "kernel": "nmf_vm_query_bwd_segments (k_vm_bwd_brick + binning)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "launches": k_launches,
                         "avg_launch_ms": k_ms / max(k_launches, 1),
                         "bytes_per_sample": {"density+normals (all samples)": BWD_BYTES_DENSITY,
                                              "appearance (bounce rows)": BWD_BYTES_APP},
                         "samples_per_step": k_samples / max(args.steps, 1),
                         "note": "algorithmic bytes assume every tap is read from HBM (SURVEY 8d); the tables are "
                                 "L2/MALL-resident and tiles accumulate in registers, so frac can exceed 1 -- compare "
                                 "`traffic` (PMC) and `mfma_frac` (dense-equivalent fp32 MFMA rate / 157.3 TFLOP/s)",
                         "mfma_frac": (k_flop / (k_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS) if k_ms > 0 else 0.0},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
