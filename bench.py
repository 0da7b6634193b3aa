#!/usr/bin/env python
"""Benchmark of the microfacet_tensorf2 hot path on MI355X (see BASELINE.json / SURVEY.md 8d).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--rays-per-gpu R] [--grid G] [--mode train|infer]

`--gpus N` with N > 1 launches itself as N ranks under torch.distributed.run (one process per GPU, RCCL); when the
driver has already done that (WORLD_SIZE set) the process is one of the ranks.

A "step" is one optimizer step (forward + backward + gradient all-reduce + Adam) over `--rays-per-gpu` rays per GPU
(default 4096 = BASELINE configs[1]; 32768 = configs[3], processed as 4096-ray chunks with gradient accumulation exactly like
train.py:509-712) of the synthetic scene S1 (solid cube in the 128^3 TensoRF grid, 512x1024 env map, 800x800 camera;
nerf_synthetic/lego is not available offline) in the STEADY-STATE phase of the reference (every secondary ray re-traced,
SURVEY F9).  Ray batches are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

`--mode infer`: a step is one full 800x800 frame rendered to completion in eval_batch_size chunks (BASELINE configs[4]).

At N = 1 the default run also reports, outside the timed region (`extras`): inference rays/s of a full frame, the training
step at the final 300^3 grid, the early phase (1000 re-traced rays), the 32768-ray step of configs[3], and the bf16-table
variant (configs[1]) next to f32.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHUNK = 4096                    # rays per forward/backward chunk (train.py `num_rays` at ~200 k primary samples)
GRID = 128
BG_RES = 512
FRAME = 800
# Algorithmic bytes per kept sample, forward (SURVEY 8d, fp32 tables, 18 taps): density value 1152 + density gradient
# 1920 + appearance 1728 = G_s = 4800.  Backward = recompute read + read-modify-write of the gradients = 3 x forward.
# The density / normal walk runs over all kept samples, the appearance walk only over the bounce rows (sparse appearance).
G_DENSITY, G_APP = 1152 + 1920, 1728
BWD_BYTES_DENSITY, BWD_BYTES_APP = 3 * G_DENSITY, 3 * G_APP
HBM_PEAK_GBS = 8000.0
# The dominant kernel (k_vm_bwd_brick) performs the scatter-add of the table gradients on the matrix cores.  Per group of 4
# samples and per plane/line pair it issues NRB row blocks x (value + 2 derivative taps) + 2 line tiles (density), or
# NRB x 2 channel halves + 2 line tiles (appearance; + 4 for the basis matrix once per group) v_mfma_f32_16x16x4_f32 of
# 16*16*4*2 = 2048 FLOP each.  NRB = ceil((BR+1)^2 / 16) = 2 for the 4^3 bricks of vm.hip (it was 6 with the 8^3 bricks
# of the r02_b profile: a third of the matrix instructions per sample now, so `achieved` counts ISSUED FLOP and fell with
# them while the launch got 1.5x faster).  f32-input MFMA peak = 157.3 TFLOP/s dense (MI355X_MICROARCH.md).
VM_BWD_NRB = 2
MFMA_FLOP_DENSITY = 3 * (3 * VM_BWD_NRB + 2) * 2048 / 4
# value-only walk (the re-traced samples: sparse normals, nmf_amd/fast_step.py): NRB value blocks + 1 line tile, and a third
# of the density-table bytes of SURVEY 8(d) (16 of the 48 packed floats per tap, 16 of 32 per line tap)
MFMA_FLOP_VALUE = 3 * (VM_BWD_NRB + 1) * 2048 / 4
BWD_BYTES_VALUE = 3 * (1152 + 1920) // 3
MFMA_FLOP_APP = (3 * (2 * VM_BWD_NRB + 2) + 4) * 2048 / 4
MFMA_F32_PEAK_TFLOPS = 157.3


def build(device, grid=None, table_dtype="f32"):
    import torch  # noqa: F401
    from nmf_amd import synthetic
    from nmf_amd.config import build_model, resolved_config
    grid = GRID if grid is None else grid
    nerf, cfg = build_model(grid=grid, bg_resolution=BG_RES, device=device)
    nerf.load_state_dict(synthetic.state_dict_s1(grid=grid, bg_resolution=BG_RES, seed=0), strict=False)
    nerf.train()
    nerf.sampler.update(nerf.rf, init=False)          # alpha mask from the density field (alphagrid.py:250-276)
    nerf.sampler.update(nerf.rf, init=True)
    nerf.model.detach_N = False                        # state after the first check_schedule (microfacet.py:117)
    nerf.model.max_retrace_rays = [nerf.model.max_brdf_rays[0]]   # steady state: all secondary rays re-traced
    if table_dtype != "f32":
        nerf.rf.set_table_dtype(table_dtype)
    return nerf, resolved_config()["params"]


class KernelTimer:
    """HIP events around every nmf_vm_query_bwd / nmf_vm_query_bwd_segments call (issued on torch's current stream, which is
    the stream the C ABI launches on) -> per-launch duration of the dominant kernel inside the timed region."""

    def __init__(self):
        import torch
        from nmf_amd import hip, functional
        self.torch = torch
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
            self.records.append((s, e, int(xyzt.shape[0]), (2 if a[9] is not None else 1) if dens else 0, app))
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
            dens = (2 if any(sg[5] is not None for sg in segs) else 1) if dens else 0      # 2: with the normal adjoint
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

    def summary(self, kind=None):
        """-> total ms, algorithmic bytes, issued MFMA flop, samples, launches of the walks of one kind (1: value-only density
        walk, 2: density walk with the normal adjoint, None: all calls incl. appearance)"""
        all_records = self.records
        if kind is not None:
            self.records = [r for r in all_records if r[3] == kind and not r[4]]
        try:
            return self._summary()
        finally:
            self.records = all_records

    def _summary(self):
        ms = sum(r[0].elapsed_time(r[1]) for r in self.records)
        nbytes = sum(m * ((BWD_BYTES_DENSITY if d == 2 else BWD_BYTES_VALUE if d == 1 else 0) + BWD_BYTES_APP * a)
                     for _, _, m, d, a in self.records)
        flop = sum(m * ((MFMA_FLOP_DENSITY if d == 2 else MFMA_FLOP_VALUE if d == 1 else 0) + MFMA_FLOP_APP * a)
                   for _, _, m, d, a in self.records)
        return ms, nbytes, flop, sum(r[2] for r in self.records), len(self.records)


def physical_cores():
    """number of physical cores this process may run on (SMT siblings counted once)"""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    cores = set()
    for c in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        cores.add(sib)
    return max(len(cores), 1)


def cpu_baseline(budget_s=150.0):
    """The CPU oracle (validated against the reference, tests/test_oracle_golden.py) timed on the host cores with the
    SURVEY 8(d) protocol: the SAME workload as the GPU step (S1, 128^3, B = 4096 rays, forward + backward of the training
    loss), one thread per physical core, one warm-up step and up to three timed steps in each phase -- early (1000 rays
    re-traced, the first 19 chunks after every (re)start) and steady state (all re-traced, what `value` is quoted on) --
    bounded to ~`budget_s` seconds of CPU work: a phase stops timing once its share of the budget is spent."""
    import torch
    from nmf_amd import synthetic
    from oracle import nmf_oracle as O
    cores = physical_cores()
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(cores)
    sd = synthetic.state_dict_s1(grid=GRID, bg_resolution=BG_RES, seed=0)
    for k, v in sd.items():
        if k != "model.brdf_sampler.angs":
            v.requires_grad_(True)
    vol = O.dense_alpha_mask({k: v.detach() for k, v in sd.items()}, O.Cfg(grid=GRID))
    out = {}
    for phase, retrace, share in (("early", 1000, 0.25), ("steady", 650000, 0.75)):
        cfg = O.Cfg(grid=GRID, detach_N=False, max_retrace_rays=(retrace,))
        times, n_samples = [], None
        t_phase = time.time()
        for i in range(4):                                  # step 0 = warm-up
            rays, focal = synthetic.camera_rays(CHUNK, seed=500 + i)
            gt = torch.rand(CHUNK, 3, generator=torch.Generator().manual_seed(i))
            for v in sd.values():
                v.grad = None
            torch.manual_seed(i)
            t0 = time.time()
            ims, st = O.render(sd, cfg, rays, focal, vol, O.Noise(), is_train=True, bg_col=torch.ones(3))
            total, _ = O.training_loss(ims, st, gt, CHUNK, sd)
            total.backward()
            dt = time.time() - t0
            n_samples = [int(x) for x in st["n_samples"]]
            if i > 0:
                times.append(dt)
            if i >= 1 and time.time() - t_phase + dt > share * budget_s:
                break
        out[phase] = dict(rays_per_s=CHUNK / (sum(times) / len(times)), s_per_step=sum(times) / len(times),
                          timed_steps=len(times), warmup_steps=1, n_samples=n_samples)
    torch.set_num_threads(prev_threads)
    st = out["steady"]
    return dict(value=st["rays_per_s"], unit="rays/s", cores=cores, kind="port",
                sample=f"B={CHUNK} rays of S1 at 128^3, forward+backward of the training loss, steady state "
                       f"(samples {st['n_samples']}): 1 warm-up + {st['timed_steps']} timed steps, {st['s_per_step']:.1f} s "
                       f"each, {cores} threads (physical cores); early phase beside it",
                early_phase=out["early"], steady_state=st)


def counters_summary():
    """Counter-backed per-kernel figures written by tools/profile_round.sh + tools/roofline_metrics.py at the profiled commit
    (rocprofv3 --pmc passes, one counter group per run): profiles/<tag>_roofline.json.  The `roofline` object of the bench line
    is computed from LIVE timings; these counters are reported beside it (`per_kernel`, `traffic`) with the tag / commit they
    were taken at, so a reader can see that the live MFMA rate and the profiled SQ_VALU_MFMA_BUSY_CYCLES agree."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_roofline.json")))
    if not files:
        return None
    try:
        return json.load(open(files[-1]))
    except (OSError, ValueError):
        return None


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, RCCL over xGMI)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def time_train(trainer, batches, focal, noise, warmup, steps, chunk, sync, global_rays=None):
    import torch  # noqa: F401
    for i in range(warmup):
        trainer.step(*batches[i % len(batches)], focal, noise=noise, update_controllers=False, fixed_chunk=chunk,
                     global_rays=global_rays)
    sync()
    t0 = time.perf_counter()
    rays_done, last, comm, first = 0, None, [], None
    for i in range(warmup, warmup + steps):
        last = trainer.step(*batches[i % len(batches)], focal, noise=noise, update_controllers=False, fixed_chunk=chunk,
                            global_rays=global_rays)
        first = first if first is not None else last
        rays_done += last["rays"]
        if last["comm_bytes"]:
            comm.append(last)
    sync()
    dt = time.perf_counter() - t0
    comm_ms = [c["comm_ms"] for c in comm[-50:]] if comm else []
    last["first_n_samples"] = first["n_samples"]
    return dt, rays_done, last, (sum(comm_ms) / len(comm_ms) if comm_ms else None)


def make_batches(nerf, n, rays_per_gpu, rank, device, distinct=48):
    """Disjoint random pixels per rank and step, resident in HBM (at most `distinct` different batches, then reused).
    The target colours are the scene's own render of those rays (training-mode forward, other noise): the loss is then the
    Monte-Carlo noise of the estimator, the parameters stay where they are and the workload (samples per step) is
    stationary over hundreds of optimizer steps -- random targets would reshape the scene within ~100 steps."""
    import torch
    from nmf_amd import synthetic
    from nmf_amd.noise import DeviceNoise
    out, focal = [], None
    gt_noise = DeviceNoise(device, seed=4242 + rank)
    white = torch.ones(3, device=device)
    for i in range(min(n, distinct)):
        rays, focal = synthetic.camera_rays(rays_per_gpu, seed=10007 * (rank + 1) + i)
        rays = rays.to(device)
        gt = torch.ones(rays_per_gpu, 3, device=device)
        with torch.no_grad():
            for s in range(0, rays_per_gpu, CHUNK):
                ims, _ = nerf(rays[s:s + CHUNK], focal, bg_col=white, is_train=True, ndc_ray=False, noise=gt_noise)
                k = ims["rgb_map"].shape[0]
                gt[s:s + k] = ims["rgb_map"].clip(0, 1)
        out.append((rays, gt))
    return out, focal


def time_infer(nerf, device, frames, warm_chunks=8):
    """full FRAME x FRAME frames of the S1 camera, eval_batch_size rays per chunk, rendered to completion"""
    import torch
    from nmf_amd import synthetic
    from nmf_amd.noise import DeviceNoise
    from nmf_amd.renderer import render_images
    rays, focal = synthetic.camera_rays(0, all_pixels=True, wh=FRAME)
    rays = rays.to(device)
    noise = DeviceNoise(device, seed=11)
    was_training = nerf.training
    nerf.eval()
    chunk = nerf.eval_batch_size
    render_images(nerf, rays[: warm_chunks * chunk], focal, chunk, noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        rgb = render_images(nerf, rays, focal, chunk, noise)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nerf.train(was_training)
    assert rgb.shape[0] == rays.shape[0]
    return dt, rays.shape[0] * frames, chunk


def extras(device, params, focal):
    """Driver-visible side measurements (N = 1 only, outside the timed region of `value`)."""
    import torch
    from nmf_amd.noise import DeviceNoise
    from nmf_amd.trainer import Trainer
    out = {}

    def sync():
        torch.cuda.synchronize()

    def train_ms(nerf, rays_per_gpu, steps, warmup):
        tr = Trainer(nerf, params)
        batches, f = make_batches(nerf, steps + warmup, rays_per_gpu, 0, device, distinct=12)
        dt, rays_done, last, _ = time_train(tr, batches, f, DeviceNoise(device, seed=5), warmup, steps, CHUNK, sync)
        return dict(ms_per_step=1e3 * dt / steps, rays_per_s=rays_done / dt, samples_per_chunk=last["n_samples"],
                    samples_per_chunk_first_step=last["first_n_samples"],
                    steps=steps, rays_per_step=rays_per_gpu)

    nerf, _ = build(device)
    dt, n, chunk = time_infer(nerf, device, frames=2)
    out["inference"] = dict(rays_per_s=n / dt, s_per_frame=dt / 2, frame=f"{FRAME}x{FRAME}", chunk=chunk,
                            note="eval mode, render to completion (renderer.py:56-106), rgb/acc outputs, S1 at 128^3")
    nerf.model.max_retrace_rays = [1000]
    out["early_phase"] = train_ms(nerf, CHUNK, 40, 10)
    out["early_phase"]["note"] = "max_retrace_rays = 1000 (first 19 chunks after every (re)start, SURVEY F9)"
    nerf.model.max_retrace_rays = [nerf.model.max_brdf_rays[0]]
    out["rays_32768_per_gpu"] = train_ms(nerf, 32768, 8, 2)
    out["rays_32768_per_gpu"]["note"] = "BASELINE configs[3] per-GPU workload: 8 chunks of 4096 rays, one optimizer step"
    del nerf
    torch.cuda.empty_cache()
    nerf300, _ = build(device, grid=300)
    out["grid_300"] = train_ms(nerf300, CHUNK, 30, 8)
    out["grid_300"]["note"] = "final grid of the schedule: 300^3, 1036 steps per ray, 41 MB of factor tables"
    del nerf300
    torch.cuda.empty_cache()
    try:
        nerf16, _ = build(device, table_dtype="bf16")
    except (AttributeError, NotImplementedError) as e:
        out["bf16_tables"] = dict(error=str(e))
    else:
        out["bf16_tables"] = train_ms(nerf16, CHUNK, 40, 10)
        out["bf16_tables"]["note"] = ("BASELINE configs[1]: factor tables read as bf16 (fp32 master copy for Adam, fp32 "
                                      "accumulation); PSNR delta in DESIGN.md")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays-per-gpu", type=int, default=CHUNK)
    ap.add_argument("--grid", type=int, default=GRID)
    ap.add_argument("--mode", choices=("train", "infer"), default="train")
    ap.add_argument("--table-dtype", choices=("f32", "bf16"), default="f32")
    ap.add_argument("--retrace", type=int, default=None,
                    help="max_retrace_rays (default: every secondary ray = the steady state; 1000 = the early phase)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    # NMF_BENCH_SHARE_GPU=1: functional test of the multi-process path on a 1-GPU box (all ranks on cuda:0, gloo because
    # RCCL refuses two ranks on one device); the driver's real runs use one GPU per rank over RCCL/xGMI.
    share = os.environ.get("NMF_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} needs {world} visible devices, found {torch.cuda.device_count()}")
    backend = os.environ.get("NMF_BENCH_BACKEND", "gloo" if share else "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # NMF_BENCH_BACKEND=nccl at --gpus 1: a process group of ONE rank over RCCL, and the gradient all-reduce entered anyway
    # (the sum over one rank is the identity).  What a 1-GPU box can show of the multi-GPU path: the library loads, the
    # collective is ordered correctly against the training pass's streams, and what pack + all-reduce + unpack cost per step.
    single_rank_comm = world == 1 and os.environ.get("NMF_BENCH_BACKEND") == "nccl"
    if single_rank_comm:
        os.environ["NMF_ALLREDUCE_SINGLE_RANK"] = "1"
        if "MASTER_ADDR" not in os.environ:
            s_ = socket.socket()
            s_.bind(("127.0.0.1", 0))
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s_.getsockname()[1]))
            s_.close()
    if world > 1 or single_rank_comm:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device, rank=rank, world_size=world)
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
    from nmf_amd.noise import DeviceNoise
    from nmf_amd.trainer import Trainer

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    torch.manual_seed(20211200)
    nerf, params = build(device, grid=args.grid, table_dtype=args.table_dtype)
    if args.retrace is not None:
        nerf.model.max_retrace_rays = [args.retrace]
    timer = KernelTimer()
    workload = (f"S1 solid-cube scene, TensoRF {args.grid}^3 (16+24 comps, {args.table_dtype} tables), env 512x1024, "
                f"800x800 camera")

    if args.mode == "infer":
        dt, n_rays, chunk = time_infer(nerf, device, frames=max(args.steps, 1))
        tt = torch.tensor([dt, float(n_rays)], dtype=torch.float64, device=device)
        if world > 1:
            tmax = tt.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            dt, n_rays = float(tmax[0]), float(tt[1])
        if rank == 0:
            print(json.dumps({
                "metric": "inference rays/sec (microfacet_tensorf2, full 800x800 frame, render to completion)",
                "value": n_rays / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": 0,
                "ms_per_step": 1e3 * dt / max(args.steps, 1), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": workload + f", eval mode, {chunk} rays per chunk; stands in for BASELINE configs[4]",
                           "parallelism": f"replicas x{world}" if world > 1 else "dp1"}}))
        if world > 1:
            dist.destroy_process_group()
        return

    trainer = Trainer(nerf, params, world_size=world, rank=rank)
    noise = DeviceNoise(device, seed=1000 + rank)
    batches, focal = make_batches(nerf, args.warmup + args.steps, args.rays_per_gpu, rank, device)
    timer.enabled = False
    for i in range(args.warmup):
        trainer.step(*batches[i % len(batches)], focal, noise=noise, update_controllers=False, fixed_chunk=CHUNK)
    sync()
    timer.enabled = True
    dt, rays_done, last, comm_ms = time_train(trainer, batches, focal, noise, 0, args.steps, CHUNK, sync)
    timer.enabled = False

    tt = torch.tensor([dt, float(rays_done), 1.0], dtype=torch.float64, device=device)
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt_max, rays_all, ranks_seen = float(tmax[0]), float(tt[1]), int(round(float(tt[2])))
    else:
        dt_max, rays_all, ranks_seen = dt, float(rays_done), 1

    chunks_per_step = -(-args.rays_per_gpu // CHUNK)
    if min(timer.rebuilds.values()) < args.steps:
        raise SystemExit(f"derived tables were not rebuilt every step: {timer.rebuilds} for {args.steps} steps")
    if rank == 0:
        # the largest field walk of the step: the value-only walk of the re-traced samples (sparse normals), else the full one
        walk_kind = 1 if any(r[3] == 1 and not r[4] for r in timer.records) else 2
        walk_key = "k_vm_bwd_density<value>" if walk_kind == 1 else "k_vm_bwd_density<normal>"
        k_ms, k_bytes, k_flop, k_samples, k_launches = timer.summary(walk_kind)
        avg_ms = k_ms / max(k_launches, 1)
        alg_gbs = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        mfma_tflops = k_flop / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        ctr = counters_summary()
        traffic, per_kernel, ctr_meta = None, None, None
        if ctr is not None:
            ks = ctr.get("kernels", {})
            traffic = ks.get(walk_key, ks.get("k_vm_bwd_brick<density>", {})).get("hbm_bytes_per_launch")
            # counter-derived figures of every kernel of the step (bound = the largest of mfma_busy / valu_busy / l2 / hbm)
            per_kernel = {k: dict(v.get("derived", {}), avg_launch_us=v.get("avg_launch_us")) for k, v in ks.items()
                          if "derived" in v}
            ctr_meta = {k: ctr.get(k) for k in ("tag", "commit", "command")}
        out = {
            "metric": "train rays/sec (microfacet_tensorf2, 4096-ray chunks, steady state)",
            "value": rays_all / dt_max, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.table_dtype == "f32" else "bf16 tables / f32 arithmetic",
            "data": "synthetic",
            "config": {"workload": workload + f", {args.rays_per_gpu} rays/GPU/step in {chunks_per_step} chunk(s) of {CHUNK}, "
                                   "fwd+bwd+all-reduce+Adam, all secondary rays re-traced (steady state); stands in for "
                                   f"BASELINE configs[{1 if args.rays_per_gpu == CHUNK else 3}] (lego / ship are not available "
                                   "offline)",
                       "rays_per_gpu": args.rays_per_gpu, "chunks_per_step": chunks_per_step, "grid": args.grid,
                       "samples_per_chunk": last["n_samples"], "samples_per_chunk_first_step": last["first_n_samples"],
                       "table_rebuilds_in_timed_region": dict(timer.rebuilds),
                       "parallelism": f"dp{world}", "ranks_seen": ranks_seen,
                       "backend": backend if (world > 1 or single_rank_comm) else None,
                       "comm_ms_per_step": comm_ms, "comm_bytes_per_step": last["comm_bytes"]},
            # Dominant kernel: the backward walk of the VM field (table-gradient scatter-add on the matrix cores).  Its
            # ceiling is the SIMD's ALU pipe: on gfx950 an fp32 MFMA and the other VALU instructions do not overlap
            # (tools/ub/coexec.hip), so MFMA cycles and VALU cycles add up.
            #   `achieved` / `frac`: ISSUED v_mfma_f32_16x16x4_f32 FLOP of the launches of the timed region / their HIP-event
            #         time (events on the launch stream), against the 157.3 TFLOP/s fp32 MFMA peak = live MFMA-busy share
            #   `alu_busy_counters`: MFMA-busy + VALU-issue share of the density walk from the committed counter run
            #         (tools/roofline_metrics.py) -- the fraction of the kernel's real ceiling
            #   `algorithmic_over_hbm`: SURVEY 8(d) bytes / time against 8 TB/s (can exceed 1: the 7.5 MB of factor tables
            #         live in L2/MALL and tiles accumulate in registers); `traffic` = fabric bytes per launch from PMC
            "roofline": {"bound": "mfma",
                         "kernel": "nmf_vm_query_bwd_segments, " + ("value-only walk of the re-traced samples" if walk_kind == 1
                                                                    else "density walk with normals") +
                                   " (binning + " + walk_key + "; runs next to the shading backward on a side stream)",
                         "achieved": mfma_tflops, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": mfma_tflops / MFMA_F32_PEAK_TFLOPS, "traffic": traffic,
                         "launches": k_launches, "avg_launch_ms": avg_ms,
                         "samples_per_launch": k_samples / max(k_launches, 1),
                         "algorithmic_bytes_per_sample": {"density value (re-traced samples)": BWD_BYTES_VALUE,
                                                          "density+normals (primary samples, bounce rows)": BWD_BYTES_DENSITY,
                                                          "appearance (bounce rows)": BWD_BYTES_APP},
                         "algorithmic_GBps": alg_gbs, "algorithmic_over_hbm": alg_gbs / HBM_PEAK_GBS,
                         "hbm_frac_counters": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and avg_ms > 0 else None,
                         "alu_busy_counters": (per_kernel or {}).get(walk_key, (per_kernel or {}).get("k_vm_bwd_brick<density>", {})).get("alu_busy"),
                         "counters": ctr_meta, "per_kernel": per_kernel},
        }
        if world == 1 and not args.no_extras and args.rays_per_gpu == CHUNK and args.grid == GRID \
                and args.table_dtype == "f32":
            del trainer
            out["extras"] = extras(device, params, focal)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1 or single_rank_comm:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
