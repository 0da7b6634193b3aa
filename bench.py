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
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # as nmf_amd/__init__.py: before the first HIP call of the process

CHUNK = 4096                    # rays per forward/backward chunk (train.py `num_rays` at ~200 k primary samples)
GRID = 128
BG_RES = 512
FRAME = 800
# SURVEY 8(d)'s byte figures per kept sample, forward (fp32 tables, 18 taps): density value 1152 + density gradient 1920 +
# appearance 1728 = G_s = 4800; fwd + bwd = 4 x forward.  The 7.5 MB of tables are L2 / MALL resident, so these bytes never
# bounded anything (round 4: 1.8 x the HBM peak at step level): they are reported as `survey_8d_over_hbm` for the record only;
# the fractions of the `roofline` object are the useful-work models of kernel_models() below.
G_DENSITY, G_APP = 1152 + 1920, 1728
HBM_PEAK_GBS = 8000.0

STARTUP_STEPS = 60          # untimed steps a process runs before its timed region in total (start-up + --warmup): see main()


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


class RebuildCounter:
    """per-step derived tables must be rebuilt after every optimizer update (packed density planes, SAT): count the rebuilds
    inside the timed region so a stale cache (= skipped work) shows up in the report"""

    def __init__(self):
        from nmf_amd import hip
        self.enabled = False
        self.rebuilds = {"vm_pack_density": 0, "sat_build": 0}
        for name in self.rebuilds:
            fn = getattr(hip, name)

            def counted(*a, _fn=fn, _name=name, **k):
                if self.enabled:
                    self.rebuilds[_name] += 1
                return _fn(*a, **k)

            setattr(hip, name, counted)


# ---- per-kernel roofline models ------------------------------------------------------------------------------------------
# Every kernel launch of libnmf_hip.so is timed with HIP events on the stream it is launched on (nmf_set_launch_probe -> the
# kernel timer of csrc/host_ext.cpp); names are the kernels' own (what rocprofv3 --kernel-trace prints).  A kernel with a
# model gets frac = USEFUL work / its own duration / peak, with the work stated per unit below and in DESIGN.md section 4, so that
# every fraction can be recomputed by hand from `sizes_per_step` and profiles/<tag>_steady_state_per_step.csv and is <= 1 by
# construction (useful <= issued <= peak x time).  The arithmetic of the path is fp32; MI355X's fp32 vector rate and its
# fp32-input matrix rate are the same 157.3 TFLOP/s (MI355X_MICROARCH.md), so that ONE peak prices VALU, MFMA and mixed
# kernels alike: `bound` = "mfma" for the kernels whose useful FLOP run on the matrix cores, "valu" otherwise.
#   BRDF MLP          SURVEY 8(d): 17 152 FLOP per secondary ray forward (66x64 + 64x64 + 64x4 multiply-adds), twice that
#                     backward (adjoint + weight-gradient products).  The kernels ISSUE 108 / 186 v_mfma_f32_32x32x16_bf16 per
#                     32 rays (split-bf16 operands): `issued_bf16_frac` prices those against the 2.5 PFLOP/s dense bf16 peak.
#   field queries     a "tap-channel" = one table entry met by one interpolation weight.  density value: 3 x (4 x 16 plane +
#                     2 x 16 line) = 288; density gradient (normals): 3 x (2 x 4 x 16 + 2 x 16) = 480 more (two derivative planes,
#                     one derivative line per pair); appearance: 3 x (4 x 24 + 2 x 24) = 432 and the 72 x 24 basis matrix.
#                     forward  = 1 FMA per tap-channel + 1 multiply per (pair, channel, product term)
#                     backward = forward recompute + 1 scatter FMA per tap-channel + 1 multiply per product term
#   marcher           20 lane-operations per candidate step (Philox4x32-10 shared by 4 steps: 12, running sum 1, position 3,
#                     coarse occupancy test 4) x rays x N steps -- the reference evaluates every one of them -- against the
#                     VALU issue peak 256 CU x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 T lane-ops/s
#   env-map adjoint   48 table updates per lookup (4 corners x 4 texels x 3 channels) against the 156 G lane-atomics/s of the
#                     memory-side units (tools/ub/atom2.hip)
#   byte-bound kernels   their compulsory bytes against 8 TB/s (Adam: 28 B per parameter; compositing, segment sums: the
#                     arrays they read and write once)
MLP_FWD_FLOP = 2 * 8576
FP32_PEAK, BF16_PEAK, HBM_PEAK, LANE_OP_PEAK, ATOMIC_PEAK = 157.3e12, 2500.0e12, 8000.0e9, 256 * 4 * 32 * 2.4e9, 156e9
MFMA_F32_PEAK_TFLOPS, MFMA_BF16_PEAK_TFLOPS = 157.3, 2500.0
# csrc/brdf_mlp.hip: v_mfma_f32_32x32x16_bf16 (2 x 32 x 32 x 16 FLOP each) issued per 32-ray tile
MLP_FWD_ISSUED_FLOP = 108 * 32768 / 32
MLP_BWD_ISSUED_FLOP = 150 * 32768 / 32      # (R5: 186 before the transposes moved from selector-matrix products into the LDS reads)
ADAM_BYTES_PER_PARAM = 28
TAPCH_VALUE, TAPCH_GRAD, TAPCH_APP = 288, 480, 432
FWD_VALUE = 2 * TAPCH_VALUE + 48                          # FLOP per sample: density value
FWD_GRAD = 2 * TAPCH_GRAD + 3 * 48                        # ... the gradient of the density on top of it (three product terms per pair)
FWD_APP = 2 * TAPCH_APP + 72 + 2 * 72 * 24                # ... appearance features incl. basis_mat
BWD_VALUE, BWD_GRAD, BWD_APP = 2 * FWD_VALUE, 2 * FWD_GRAD, 2 * FWD_APP
MARCH_LANE_OPS = 20
L2_PEAK_GBS = 34500.0


def kernel_models(sz, n_params):
    """sz: sizes of one step (rays B, samples M0 M1, secondary rays R0 R1, bounce rows Mb0 Mb1, N steps per ray)
    -> {kernel name: (bound, useful work per STEP, peak, unit)}; names as libnmf_hip.so launches them"""
    B, M0, M1, R0, R1, Mb0, Mb1, N = (sz[k] for k in ("B", "M0", "M1", "R0", "R1", "Mb0", "Mb1", "N"))
    R = R0 + R1
    F, H, A = "TFLOP/s", "GB/s", "G updates/s"
    m = {
        "k_brdf_mlp_bwd": ("mfma", 2.0 * MLP_FWD_FLOP * R, FP32_PEAK, F),
        "k_brdf_mlp_fwd": ("mfma", 1.0 * MLP_FWD_FLOP * R, FP32_PEAK, F),
        # backward walks: value-only over the re-traced samples; value + gradient over the primary samples and the bounce rows
        # of the re-traced level (their normals); appearance over the bounce rows of both levels
        "k_vm_bwd_density<false>": ("mfma", float(BWD_VALUE * M1), FP32_PEAK, F),
        "k_vm_bwd_density<true>": ("mfma", float((BWD_VALUE + BWD_GRAD) * (M0 + Mb1)), FP32_PEAK, F),
        "k_vm_bwd_brick<false, 1>": ("mfma", float(BWD_APP * (Mb0 + Mb1)), FP32_PEAK, F),
        "k_vm_fwd<float>": ("valu", float((FWD_VALUE + FWD_GRAD) * M0), FP32_PEAK, F),
        "k_vm_sigma<float>": ("valu", float(FWD_VALUE * M1), FP32_PEAK, F),
        "k_vm_rows_dn<float>": ("valu", float((FWD_VALUE + FWD_GRAD) * Mb1), FP32_PEAK, F),
        "k_vm_app_rows<float>": ("valu", float(FWD_APP * (Mb0 + Mb1)), FP32_PEAK, F),
        "k_march_count16": ("valu", float(MARCH_LANE_OPS) * R0 * N, LANE_OP_PEAK, "T lane-ops/s"),
        "k_march_fill16": ("valu", float(MARCH_LANE_OPS) * R0 * N, LANE_OP_PEAK, "T lane-ops/s"),
        "k_march_count": ("valu", float(MARCH_LANE_OPS) * B * N, LANE_OP_PEAK, "T lane-ops/s"),
        "k_march_fill": ("valu", float(MARCH_LANE_OPS) * B * N, LANE_OP_PEAK, "T lane-ops/s"),
        # (the binned env-map adjoint is three kernels over the same lookups: priced as a group, KERNEL_GROUPS)
        "group:env_adjoint": ("atomics", 48.0 * R, ATOMIC_PEAK, A),
        "group:brdf_mlp_backward": ("mfma", 2.0 * MLP_FWD_FLOP * R, FP32_PEAK, F),
        "k_env_lookup_fwd<1>": ("hbm", 192.0 * (R + 5000), HBM_PEAK, H),
        "k_adam": ("hbm", float(ADAM_BYTES_PER_PARAM * n_params), HBM_PEAK, H),
        "k_composite_bwd_wave<16>": ("hbm", 20.0 * M1, HBM_PEAK, H),
        "k_composite_fwd_wave<8>": ("hbm", 16.0 * M1, HBM_PEAK, H),
        "k_plan_hist": ("hbm", 20.0 * (M0 + M1 + 2 * Mb1 + Mb0), HBM_PEAK, H),
        "k_place_records<false>": ("hbm", 56.0 * (M0 + M1 + Mb1), HBM_PEAK, H),
    }
    return m


# kernels that are one operation of the pass (the numbers VERDICT r04 tracks): summed rows "group:<name>" in per_kernel
KERNEL_GROUPS = {
    "brdf_mlp_backward": ("k_brdf_mlp_bwd", "k_brdf_mlp_reduce"),
    "brick_sort": ("k_plan_hist", "k_bins_scan", "k_place_records<false>", "k_place_records<true>", "k_bins_partial", "k_bins_final",
                   "k_plan_place", "k_brick_records<false>", "k_brick_records<true>"),
    "env_adjoint": ("k_env_bin_count<1>", "k_env_bin_scatter<1>", "k_env_bin_accum<1>", "k_env_lookup_bwd<1>"),
    "field_walks": ("k_vm_bwd_density<false>", "k_vm_bwd_density<true>", "k_vm_bwd_brick<false, 1>", "k_vm_bwd_brick<false, 2>",
                    "k_vm_bwd_brick<true, 2>"),
    "marcher": ("k_march_count16", "k_march_fill16", "k_march_count", "k_march_fill", "k_scan_fused", "k_scan_partial"),
}


def issued_beside(name, sz):
    """what a kernel issues next to what it usefully computes (per step)"""
    R = sz["R0"] + sz["R1"]
    if name == "k_brdf_mlp_bwd":
        return dict(issued_bf16_flop=MLP_BWD_ISSUED_FLOP * R, issued_peak=BF16_PEAK)
    if name == "k_brdf_mlp_fwd":
        return dict(issued_bf16_flop=MLP_FWD_ISSUED_FLOP * R, issued_peak=BF16_PEAK)
    return None


# rocprofv3 prints template kernels as "void k_name<args>(...)"; the counter files of tools/roofline_metrics.py key them without
# the template arguments except for the two walks
def counter_key(name):
    if name == "k_vm_bwd_density<false>":
        return "k_vm_bwd_density<value>"
    if name == "k_vm_bwd_density<true>":
        return "k_vm_bwd_density<normal>"
    if name.startswith("k_vm_bwd_brick<false, 1>"):
        return "k_vm_bwd_brick<appearance>"
    return name.split("<")[0]


def per_kernel_table(timing, steps, sz, n_params, counters):
    """timing: {kernel: (ms, launches)} over `steps` instrumented steps (keys starting with '@' are per-stream sums) -> rows sorted
    by time"""
    models = kernel_models(sz, n_params)
    ks = (counters or {}).get("kernels", {})
    rows = {}
    timing = dict(timing)
    for gname, members in KERNEL_GROUPS.items():
        got = [timing[k] for k in members if k in timing]
        if got:
            timing["group:" + gname] = (sum(g[0] for g in got), sum(g[1] for g in got))
    for name, (ms, calls) in sorted(((k, v) for k, v in timing.items() if not k.startswith("@")), key=lambda kv: -kv[1][0]):
        us = 1e3 * ms / steps
        bound, work, peak, unit = models.get(name, ("latency", None, None, None))
        rec = {"us_per_step": round(us, 1), "launches_per_step": round(calls / steps, 2), "bound": bound}
        if work is not None and us > 0:
            rec["useful_per_step"] = work
            rec["achieved"] = work / (us * 1e-6)
            rec["peak"] = peak
            rec["unit_base"] = unit
            rec["frac"] = round(rec["achieved"] / peak, 4)
            extra = issued_beside(name, sz)
            if extra:
                rec["issued_bf16_frac"] = round(extra["issued_bf16_flop"] / (us * 1e-6) / extra["issued_peak"], 4)
        else:
            rec["frac"] = None
        d = ks.get(counter_key(name), {}).get("derived") or ks.get(name, {}).get("derived")
        if d:
            rec["counters"] = {k: d[k] for k in ("alu_busy", "mfma_busy", "hbm_frac", "l2_frac", "waves_per_simd") if k in d}
        rows[name] = rec
    return rows


def step_bytes(sz, grid, n_params):
    """SURVEY 8(d) bytes of one optimizer step (every tap charged to HBM, dense appearance and normals on every sample) and the
    bytes the pass needs after sparse appearance / sparse normals (values for the re-traced samples, appearance + normals on the
    bounce rows only); fwd + bwd = 4 x forward (1 read + 1 recompute read + read-modify-write of the gradients)."""
    B, M0, M1, R0, R1, Mb0, Mb1 = (sz[k] for k in ("B", "M0", "M1", "R0", "R1", "Mb0", "Mb1"))
    fixed = 2 * 3 * BG_RES * 2 * BG_RES * 4 + 3 * (3 * 16 * grid * grid * 4) + ADAM_BYTES_PER_PARAM * n_params
    n_steps = sz.get("N", 440)
    survey = 4 * 4800 * (M0 + M1) + 576 * (R0 + R1 + B) + n_steps * (B + R0) + 40 * B + fixed
    needed = (4 * (M0 * G_DENSITY + M1 * 1152 + Mb1 * G_DENSITY + (Mb0 + Mb1) * G_APP) + 576 * (R0 + R1) + n_steps * (B + R0) + 40 * B
              + fixed)
    return survey, needed


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


def cpu_baseline(budget_s=150.0, warm=1, timed=3):
    """The CPU oracle (validated against the reference, tests/test_oracle_golden.py) timed on the host cores with the
    SURVEY 8(d) protocol: the SAME workload as the GPU step (S1, 128^3, B = 4096 rays, forward + backward of the training
    loss), one thread per physical core, one warm-up step and up to three timed steps in each phase -- early (1000 rays
    re-traced, the first 19 chunks after every (re)start) and steady state (all re-traced, what `value` is quoted on) --
    bounded to ~`budget_s` seconds of CPU work: a phase stops timing once its share of the budget is spent.
    `--cpu-baseline-steps 3,10` is SURVEY 8(d)'s full protocol (3 warm-up + 10 timed steps per phase, ~6 minutes of host time,
    no budget): run once per round and kept in profiles/."""
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
        for i in range(warm + timed):                       # the first `warm` steps are warm-up
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
            if i >= warm:
                times.append(dt)
            if budget_s is not None and i >= warm and time.time() - t_phase + dt > share * budget_s:
                break
        out[phase] = dict(rays_per_s=CHUNK / (sum(times) / len(times)), s_per_step=sum(times) / len(times),
                          timed_steps=len(times), warmup_steps=warm, n_samples=n_samples)
    torch.set_num_threads(prev_threads)
    st = out["steady"]
    return dict(value=st["rays_per_s"], unit="rays/s", cores=cores, kind="port",
                sample=f"B={CHUNK} rays of S1 at 128^3, forward+backward of the training loss, steady state "
                       f"(samples {st['n_samples']}): {warm} warm-up + {st['timed_steps']} timed steps, {st['s_per_step']:.1f} s "
                       f"each, {cores} threads (physical cores); early phase beside it",
                early_phase=out["early"], steady_state=st,
                protocol=f"{warm}+{timed} steps per phase within a {budget_s} s budget" if budget_s else f"{warm}+{timed} steps per phase (SURVEY 8d)",
                protocol_full="profiles/r04_cpu_baseline_3_10.json (SURVEY 8d's 3 warm-up + 10 timed steps per phase on the GPU box's host: "
                              "213 rays/s steady state, 634 rays/s early phase; `python bench.py --cpu-baseline-steps 3,10`)")


LINE_LIMIT = 6144           # bytes: the driver keeps an 8 KB tail of stdout and parses the last line out of it (round 5's 21 KB line was not parsed)
DETAIL_FILE = "bench_detail.json"


def _r(x, nd=4):
    """Numbers of the line rounded to `nd` significant digits below 1e4 (and to integers above), recursively."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{nd}g}") if abs(x) < 1e4 else round(x)
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def compact_line(out, detail_path=DETAIL_FILE):
    """The ONE JSON line of the contract, from the full result dict `out`: the contract's keys, `roofline` of the dominant kernel,
    `cpu_baseline`, `psnr_at_iter` and ONE number per extras leg.  Everything else (the per-kernel table, notes, the long legs)
    is in `detail_path`, which the line names.  Kept below LINE_LIMIT bytes (tests/test_bench_line.py)."""
    cfg = dict(out.get("config") or {})
    wl = str(cfg.get("workload", ""))
    keep_cfg = ("rays_per_gpu", "chunks_per_step", "chunks_in_flight", "grid", "samples_per_chunk", "parallelism", "ranks_seen", "backend",
                "comm_ms_per_step", "comm_bytes_per_step", "comm_exposed_ms", "host_cpu_ms_per_step", "startup_steps", "host_pass",
                "core_switches", "operator_graph_fallbacks")
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": wl if len(wl) <= 300 else wl[:297] + "...", **{k: cfg[k] for k in keep_cfg if k in cfg}}
    roof = out.get("roofline") or {}
    if "kernel" in roof:
        st = roof.get("step") or {}
        line["roofline"] = {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "issued_bf16_frac", "traffic",
                                                     "avg_launch_us", "launches", "sizes_per_step")}
        line["roofline"]["counters_of_kernel"] = roof.get("counters_of_kernel")
        line["roofline"]["step"] = {k: st.get(k) for k in ("wall_us", "main_stream_kernel_us", "critical_path_frac", "device_time_sum_us",
                                                           "survey_8d_over_hbm", "needed_over_l2")}
        groups = {k[6:]: v.get("us_per_step") for k, v in (roof.get("per_kernel") or {}).items() if k.startswith("group:")}
        if groups:
            line["roofline"]["group_us_per_step"] = groups
    elif roof:
        line["roofline"] = {"note": str(roof.get("note", ""))[:200]}
    cb = out.get("cpu_baseline")
    if cb:
        ss = cb.get("steady_state") or {}
        line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "sample": f"B={CHUNK} rays of S1 at 128^3, fwd+bwd of the training loss, steady state, "
                                          f"{ss.get('warmup_steps')}+{ss.get('timed_steps')} steps of {ss.get('s_per_step', 0):.1f} s",
                                "protocol": cb.get("protocol"),
                                "early_phase_rays_per_s": (cb.get("early_phase") or {}).get("rays_per_s")}
    ps = out.get("psnr_at_iter")
    if ps:
        line["psnr_at_iter"] = {k: ps.get(k) for k in ("test_psnr_db", "reference_mean_db", "delta_db", "delta_stderr_db", "seeds",
                                                       "reference_seeds", "pooled", "error") if k in ps}
    ex = out.get("extras")
    if ex:
        one = {}
        for k, v in ex.items():
            if k == "psnr_at_iter" or not isinstance(v, dict):
                continue
            if "error" in v:
                one[k] = "error"
            elif k == "schedule_weighted":
                one[k + "_rays_per_s"] = v.get("rays_per_s")
            elif "ms_per_step" in v and k not in ("rays_32768_per_gpu", "rays_32768_per_gpu_sequential", "rays_32768_per_gpu_budgets_x4",
                                                   "inference", "grid_300"):
                one[k + "_ms"] = v["ms_per_step"]
            else:
                one[k + "_rays_per_s"] = v.get("rays_per_s")
        line["extras"] = one
    line["detail"] = detail_path
    line = _r(line)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:                     # never an unparsable line: drop the optional objects, largest first
        for k in ("extras", "psnr_at_iter"):
            line.pop(k, None)
            text = json.dumps(line, separators=(",", ":"))
            if len(text) <= LINE_LIMIT:
                break
    return text


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


def scale_budgets(nerf, f):
    """per-chunk budgets of the reference's config times f (the chunk grows with them): sampler.max_samples, model.max_brdf_rays;
    every secondary ray stays re-traced"""
    nerf.sampler.max_samples = int(nerf.sampler.max_samples) * f
    nerf.model.max_brdf_rays = [int(v) * f for v in nerf.model.max_brdf_rays]
    nerf.model.max_retrace_rays = [nerf.model.max_brdf_rays[0]]


def flush_c_stdio():
    """fflush(NULL) + Python's own buffers: what native libraries (RCCL's banner) have queued on stdout goes out now"""
    import ctypes
    sys.stdout.flush()
    sys.stderr.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def time_train(trainer, batches, focal, noise, warmup, steps, chunk, sync, global_rays=None):
    import torch  # noqa: F401
    nb = len(batches)
    for i in range(warmup):
        trainer.step(*batches[i % nb], focal, noise=noise, update_controllers=False, fixed_chunk=chunk,
                     global_rays=global_rays)
    sync()
    t0 = time.perf_counter()
    cpu0 = time.process_time()
    rays_done, last, comm, first = 0, None, [], None
    for i in range(warmup, warmup + steps):
        # (the loop knows its next batch, as train.py's permutation sampler does: the sampler of its first chunk is prefetched)
        last = trainer.step(*batches[i % nb], focal, noise=noise, update_controllers=False, fixed_chunk=chunk,
                            global_rays=global_rays)
        first = first if first is not None else last
        rays_done += last["rays"]
        if last["comm_bytes"]:
            comm.append(last)
    sync()
    dt = time.perf_counter() - t0
    comm_ms = [c["comm_ms"] for c in comm[-50:]] if comm else []
    last["first_n_samples"] = first["n_samples"]
    # CPU time of THIS process over the timed steps (all its threads): what one rank needs of the host; 8 ranks on a node need 8 x
    last["host_cpu_ms_per_step"] = 1e3 * (time.process_time() - cpu0) / max(steps, 1)
    return dt, rays_done, last, (sum(comm_ms) / len(comm_ms) if comm_ms else None)


def reference_style_step(nerf, optimizer, rays, rgb_gt, focal, params, noise, chunk, ori_lambda=None, pred_lambda=None):
    """One optimizer step written the way the reference's own loop writes it (train.py:497-747): per chunk `nerf(...)` ->
    (images, statistics), the loss assembled with plain torch operations from rgb_map and EVERY statistic the reference reads
    (prediction / distortion / orientation terms, the three zero-weight regularisers, density_L1), `total_loss.backward()`, then
    `optimizer.step()`.  Nothing of nmf_amd.trainer: this is what a maintainer gets who swaps the operator classes in under
    train.py.  -> (rays used, n_samples of the last chunk, summed photometric loss as a tensor)"""
    import torch
    optimizer.zero_grad(set_to_none=True)
    n_total = rays.shape[0]
    lbatch = n_total
    ori_lambda = params["ori_lambda"] if ori_lambda is None else ori_lambda
    pred_lambda = params["pred_lambda"] if pred_lambda is None else pred_lambda
    bg_col = torch.ones(3, device=rays.device)
    used, n_samples, photo = 0, None, None
    for s in range(0, n_total, chunk):
        r, gt = rays[s:s + chunk], rgb_gt[s:s + chunk]
        ims, st = nerf(r, focal, bg_col=bg_col, is_train=True, ndc_ray=False, noise=noise)
        n_samples = st["n_samples"]
        if n_samples[0] == 0:
            continue
        rgb_map = ims["rgb_map"].clip(max=1)
        whole_valid = st["whole_valid"]
        kept = rgb_map.shape[0]
        assert whole_valid.shape[0] == r.shape[0]
        gt_kept = gt[:kept]                       # (= rgb_train[whole_valid]: the kept rays are a prefix, alphagrid.py:353-364)
        loss = ((rgb_map.clip(0, 1) - gt_kept.clip(0, 1)) ** 2).sum()
        total = (loss + params.get("distortion_lambda", 0) * st["distortion_loss"].sum() + ori_lambda * st["ori_loss"].sum()
                 + params.get("envmap_lambda", 0) * st["envmap_reg"].sum() + params.get("diffuse_lambda", 0) * st["diffuse_reg"].sum()
                 + params.get("brdf_lambda", 0) * st["brdf_reg"].sum() + pred_lambda * st["prediction_loss"].sum())
        if params["L1_weight_initial"] > 0:
            total = total + params["L1_weight_initial"] * nerf.rf.density_L1()
        total = total / lbatch
        total.backward()
        used += kept
        photo = loss.detach() if photo is None else photo + loss.detach()
    optimizer.step()
    return used, n_samples, photo


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


def time_infer(nerf, device, frames, warm_frames=1, chunk=None):
    """full FRAME x FRAME frames of the S1 camera, eval_batch_size rays per chunk, rendered to completion.  The warm-up is
    one whole frame: the chunks of a frame differ in their sample counts, and the first pass over them is the caching
    allocator growing its pools (tools/infer_host_profile.py: 333 ms for the first frame, 101 ms for every later one) --
    eight warm chunks, as rounds 1-3 used, left most of that inside the timed region.
    -> (seconds of the timed frames, rays rendered, chunk size, seconds of the first (cold) frame)"""
    import torch
    from nmf_amd import synthetic
    from nmf_amd.noise import DeviceNoise
    from nmf_amd.renderer import render_images
    rays, focal = synthetic.camera_rays(0, all_pixels=True, wh=FRAME)
    rays = rays.to(device)
    noise = DeviceNoise(device, seed=11)
    was_training = nerf.training
    nerf.eval()
    chunk = chunk or nerf.eval_batch_size
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(max(warm_frames, 1)):
        render_images(nerf, rays, focal, chunk, noise)
    torch.cuda.synchronize()
    cold = (time.perf_counter() - t0) / max(warm_frames, 1)
    t0 = time.perf_counter()
    for _ in range(frames):
        rgb = render_images(nerf, rays, focal, chunk, noise)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nerf.train(was_training)
    assert rgb.shape[0] == rays.shape[0]
    return dt, rays.shape[0] * frames, chunk, cold


SCHEDULE = ((128, 2000), (162, 1000), (196, 1000), (231, 1500), (265, 1500), (300, 23000))   # grid, iterations at it


def reference_psnr_seeds():
    """test PSNR [seed, evaluation] of every run of the REFERENCE's own training loop on the S2 configuration that the build container
    produced: tests/golden/psnr_trace.npz (six seeds with their initial states) + tests/golden/psnr_ref_more_*.npz (further seeds,
    PSNR only: tests/golden/make_psnr_more.py) -> (array, evaluation iterations)"""
    import glob
    import numpy as np
    g = np.load(os.path.join(ROOT, "tests", "golden", "psnr_trace.npz"))
    rows = [g[f"s{s_}/test_psnr"].mean(-1) for s_ in range(int(g["n_seeds"]))]
    # (+ round 6: the runs that also keep their trajectories, tests/golden/make_psnr_traj.py)
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "psnr_ref_more*.npz"))
                    + glob.glob(os.path.join(ROOT, "tests", "golden", "psnr_ref_traj_*.npz"))):
        try:
            with np.load(f) as z:
                rows += [z[k].mean(-1) for k in sorted(z.files) if k.endswith("/test_psnr")]
        except (OSError, ValueError, EOFError):          # (a file cut off while it was being extended: the others still count)
            continue
    return np.stack(rows), [int(v) for v in g["psnr_at"]]


def psnr_runs(device, seeds, traj=None, params_over=None, on_iter=None, sampler="reference"):
    """One 300-iteration training of the S2 configuration per seed, each from a FRESH initialisation (the constructors' random
    initial parameters under torch.manual_seed(seed), calibration as train.py:429-437), device noise, the data set of the fixture
    -> test PSNR [seed, evaluation] (8-bit formula, renderer.py:399-401), rays per second incl. the evaluations.
    traj (a list): receives one dict per seed with the run's TRAJECTORY in the layout of tests/golden/make_psnr_traj.py (per chunk: loss
    back-propagated, num_rays, rays in / kept, n_samples, max_retrace; per iteration: global batch, learning rates, gradient and
    parameter norms) -- tools/psnr_trajectory.py compares the seed means of the two sides.
    sampler: "reference" = which rays a chunk gets as train.py:34-51 decides it (nmf_amd.controllers.SimpleSampler: overlapping chunks,
    what `python -m nmf_amd.train` does in one process); "disjoint" = consecutive slices of a permutation (rounds 2-5 of this file: the
    cause of the +0.17 dB at 200 iterations)"""
    import numpy as np
    import torch
    from nmf_amd.config import build_model, resolved_config
    from nmf_amd.noise import DeviceNoise
    from nmf_amd.renderer import psnr_8bit, render_images
    from nmf_amd.trainer import Trainer
    g = np.load(os.path.join(ROOT, "tests", "golden", "psnr_trace.npz"))
    G0, BG, res = int(g["grid0"]), int(g["bg_res"]), int(g["res"])
    ov = dict(line.split("=", 1) for line in str(g["overrides"]).split("\n"))
    ints = lambda k: [int(v) for v in ov[k].strip("[]").split(",")]  # noqa: E731
    over = {"sampler.update_list": [10 ** 9], "rf.upsamp_list": [10 ** 9], "rf.N_voxel_init": G0 ** 3, "rf.N_voxel_final": G0 ** 3,
            "sampler.max_samples": ints("model.arch.sampler.max_samples")[0], "model.max_brdf_rays": ints("model.arch.model.max_brdf_rays"),
            "model.target_num_samples": ints("model.arch.model.target_num_samples"),
            "model.max_retrace_rays": ints("model.arch.model.max_retrace_rays"), "model.rays_per_ray": ints("model.arch.model.rays_per_ray")[0]}
    mn, mx, start, target = (int(v) for v in g["params_params"])
    params = dict(resolved_config()["params"], n_iters=int(ov["model.params.n_iters"]), batch_size=mn, min_batch_size=mn,
                  max_batch_size=mx, starting_batch_size=start, target_num_samples=target)
    params.update(params_over or {})              # (factor experiments of tools/psnr_trajectory.py, e.g. ori_lambda = 0)
    rays_tr, rgb_tr = torch.as_tensor(g["rays_train"]).to(device), torch.as_tensor(g["rgb_train"]).to(device)
    rays_te, rgb_te = torch.as_tensor(g["rays_test"]).to(device), torch.as_tensor(g["rgb_test"]).to(device)
    focal, n_views = float(g["focal"]), rays_te.shape[0] // (res * res)
    at = [int(v) for v in g["psnr_at"]]
    n_total = rays_tr.shape[0]
    out, t0, rays_seen = [], time.perf_counter(), 0
    for seed in seeds:
        torch.manual_seed(20211200 + 1000 + seed)              # (train.py:906: the reference seeds its constructors the same way)
        nerf, _ = build_model(grid=G0, bg_resolution=BG, device=device, overrides=over)
        nerf.train()
        nerf.sampler.update(nerf.rf, init=True)
        with torch.no_grad():                                   # train.py:429-437
            xyz = torch.rand(100000, 4, device=device) * 2 - 1
            xyz[:, 3] *= 0
            nerf.model.calibrate(None, xyz, nerf.rf.compute_appfeature(xyz), nerf.bg_module.mean_color().mean())
        tr = Trainer(nerf, params)
        noise = DeviceNoise(device, seed=5000 + seed)
        gen = torch.Generator(device=device).manual_seed(seed)
        perm, cur, row = torch.randperm(n_total, device=device, generator=gen), 0, []
        T = None
        if traj is not None:
            T = dict(chunk_num_rays=[], chunk_rays_in=[], chunk_kept=[], chunk_n_samples=[], chunk_iter=[], chunk_loss=[],
                     chunk_max_retrace=[], iter_lbatch=[], iter_lr=[], iter_max_retrace=[], iter_num_chunks=[], iter_gradnorm=[],
                     iter_param_norm=[])
            names = sorted(n_ for n_, _ in nerf.named_parameters())
            byname = dict(nerf.named_parameters())
        ref_sampler = None
        if sampler == "reference":          # train.py:34-51: one nextids() per chunk, the cursor moved before the slice (overlapping chunks)
            from nmf_amd.controllers import SimpleSampler
            ref_sampler = SimpleSampler(n_total, mn, lambda n_: torch.randperm(n_, device=device, generator=gen))

            def fetch(n_):
                ids_ = ref_sampler.nextids(n_)
                return rays_tr[ids_], rgb_tr[ids_]
        for it in range(at[-1]):
            nb = tr.lbatch_size()
            rec = [] if T is not None else None
            if T is not None:
                T["iter_lr"].append([float(g_["lr"]) for g_ in tr.optimizer.param_groups])
            if ref_sampler is not None:
                st = tr.step(None, None, focal, noise=noise, global_rays=nb, trace=rec, fetch=fetch)
            else:
                if cur + nb > n_total:
                    perm, cur = torch.randperm(n_total, device=device, generator=gen), 0
                ids = perm[cur:cur + nb]
                cur += nb
                st = tr.step(rays_tr[ids], rgb_tr[ids], focal, noise=noise, global_rays=nb, trace=rec)
            rays_seen += st["rays"]
            if on_iter is not None:                 # (tools/trained_state_dump.py: the model after a given iteration)
                on_iter(seed, it, nerf, tr)
            if T is not None:
                for r_ in rec:
                    if "total" not in r_:               # a chunk without a sample: train.py:567-568 skips it before the backward
                        continue
                    ns = list(r_["n_samples"])
                    T["chunk_num_rays"].append(r_["num_rays"]); T["chunk_rays_in"].append(r_["rays_in"]); T["chunk_kept"].append(r_["kept"])
                    T["chunk_n_samples"].append(ns + [0] * (2 - len(ns))); T["chunk_iter"].append(it)
                    T["chunk_loss"].append(float(r_["total"])); T["chunk_max_retrace"].append(int(r_["max_retrace"][0]))
                T["iter_lbatch"].append(nb); T["iter_num_chunks"].append(len(rec))
                T["iter_max_retrace"].append(int(nerf.model.max_retrace_rays[0]))
                T["iter_gradnorm"].append([float(byname[n_].grad.norm()) if byname[n_].grad is not None else float("nan") for n_ in names])
                T["iter_param_norm"].append([float(byname[n_].detach().double().norm()) for n_ in names])
            if it + 1 in at:
                nerf.eval()
                # 800 rays per evaluation chunk, as the reference runs of the fixture (make_train_trace.py test_psnr): the per-CHUNK
                # budgets (max_retrace_rays, max_brdf_rays[1]) make the rendering depend on the chunk size
                pred = render_images(nerf, rays_te, focal, 800, noise, draw_debug=True)
                nerf.train()
                pv, gv = pred.reshape(n_views, -1, 3), rgb_te.reshape(n_views, -1, 3)
                row.append(float(torch.stack([psnr_8bit(pv[i], gv[i]) for i in range(n_views)]).mean()))
        out.append(row)
        if T is not None:
            T = {k_: np.asarray(v_) for k_, v_ in T.items()}
            T.update(test_psnr=np.asarray(row), names=names, seed=seed)
            traj.append(T)
        del tr, nerf
    torch.cuda.synchronize()
    return np.asarray(out), rays_seen / (time.perf_counter() - t0), (G0, BG, res, mn)


def psnr_at_iter(device, n_seeds=128):
    """BASELINE metric, second half ("PSNR@iter") and north_star's "PSNR within 0.05 dB of reference after equal iterations": the
    300-iteration S2 orbit training (48^3, 24 views of 32 x 32, 1024-ray batches, 128 secondary rays per sample, the reference's lr
    schedule) as a comparison of two DISTRIBUTIONS over seeds -- every training is its own realisation of a stochastic optimisation
    (seed-to-seed standard deviation 0.25-0.4 dB on either side; a run paired with the reference by noise diverges at the first bounce
    count that floors the other way, tests/test_hip_timed_path.py) -- between `n_seeds` trainings here, fresh initialisations, and ALL
    runs of the reference's own loop the build container produced (reference_psnr_seeds): difference of the means and its standard
    error per evaluation.
    128 seeds (about 1 s each): 32-seed blocks of this measurement scatter by +- 0.13 dB at iteration 100 (twelve blocks of the 384-run
    measurement of round 6: 30.09 ... 30.49, seeds 0-31 give 30.16), more than a 32-seed standard error admits to."""
    import numpy as np
    if not os.path.exists(os.path.join(ROOT, "tests", "golden", "psnr_trace.npz")):
        return dict(error="tests/golden/psnr_trace.npz missing")
    ref, at = reference_psnr_seeds()
    hip, rays_per_s, (G0, BG, res, mn) = psnr_runs(device, range(n_seeds))
    se = lambda x: x.std(0, ddof=1) / np.sqrt(x.shape[0])  # noqa: E731
    delta = hip.mean(0) - ref.mean(0)
    dse = np.sqrt(se(hip) ** 2 + se(ref) ** 2)
    r3 = lambda v: {str(a_): round(float(x), 3) for a_, x in zip(at, v)}  # noqa: E731
    return dict(test_psnr_db=r3(hip.mean(0)), seed_stderr_db=r3(se(hip)), seeds=int(hip.shape[0]),
                reference_mean_db=r3(ref.mean(0)), reference_seed_stderr_db=r3(se(ref)), reference_seeds=int(ref.shape[0]),
                delta_db=r3(delta), delta_stderr_db=r3(dse),
                train_rays_per_s_incl_evals=round(rays_per_s, 1),
                pooled="384 runs here vs 123 of the reference (round 6, one box): -0.04 / +0.01 / -0.01 +- 0.05 / 0.04 / 0.04 dB, "
                       "profiles/r06_psnr_384_vs_123.txt",
                config=f"S2 orbit, TensoRF {G0}^3, env {BG}x{2 * BG}, {res}x{res} views, {mn}-ray batches, the reference's ray sampler; {hip.shape[0]} trainings here "
                       f"(fresh initialisations) against {ref.shape[0]} runs of the reference's own loop; means over seeds, delta = here - "
                       "reference with the standard error of that difference")


def host_path_legs(device, params):
    """The three legs of `extras` that are bound by the HOST: the reference's loop restated on the drop-in classes, the Trainer through
    TensorNeRF.forward + backward(), the operator graph under the reference's loop.  -> {name: dict}.  `extras` runs them in a FRESH
    process (`python bench.py --leg host_paths`), the protocol of the headline number: inside the bench process, behind its 400 steps
    and 7000 timing events, they ran 0.3-0.5 ms per step slower than on their own (round 5: 2.34 against 1.88 ms on one box)."""
    import torch
    from nmf_amd.noise import DeviceNoise
    from nmf_amd.trainer import Trainer
    out = {}

    def sync():
        torch.cuda.synchronize()

    nerf, _ = build(device)
    # (first: these legs are bound by the host, and a process that has been through the larger legs below issues more slowly)
    # what a maintainer of the reference gets who swaps the operator classes in and keeps train.py's loop:
    # reference_style_step above -- TensorNeRF.forward, the loss in torch operations, total_loss.backward(), optimizer.step() --
    # steady state, 4096 rays.  The forward / backward of a chunk enter the same C++ pass as `value` through ONE autograd node.
    def reference_loop_ms(steps, warmup, fused):
        from nmf_amd.optim import FusedAdam
        nerf.fused_training_pass = fused
        graph0 = nerf.operator_graph_forwards
        opt = FusedAdam(nerf.get_optparam_groups(), betas=tuple(params["betas"]), eps=params["eps"], weight_decay=params["weight_decay"])
        batches, f = make_batches(nerf, steps + warmup, CHUNK, 0, device, distinct=12)
        nz = DeviceNoise(device, seed=5)
        for i in range(warmup):
            reference_style_step(nerf, opt, *batches[i % len(batches)], f, params, nz, CHUNK)
        sync()
        t0 = time.perf_counter()
        rays_done = 0
        for i in range(warmup, warmup + steps):
            used, n_samples, _ = reference_style_step(nerf, opt, *batches[i % len(batches)], f, params, nz, CHUNK)
            rays_done += used
        sync()
        dt = time.perf_counter() - t0
        nerf.fused_training_pass = True
        left = nerf.operator_graph_forwards - graph0
        if fused and left:                 # (a fused leg that silently times the operator graph is not that leg)
            raise SystemExit(f"reference_loop: {left} of {steps + warmup} training forwards left the fused pass for the operator graph")
        return dict(ms_per_step=1e3 * dt / steps, rays_per_s=rays_done / dt, samples_per_chunk=n_samples, steps=steps, rays_per_step=CHUNK,
                    operator_graph_forwards=left)

    out["reference_loop"] = reference_loop_ms(60, 20, True)
    out["reference_loop"]["note"] = ("the loop of the reference's train.py:497-747 (bench.reference_style_step: TensorNeRF.forward, loss in torch "
                                     "operations incl. every statistic train.py reads, total_loss.backward(), FusedAdam.step()) on the drop-in "
                                     "operator classes: one autograd node per chunk over the C++ pass; ~40 small torch launches of loss "
                                     "arithmetic per chunk are the loop's own")
    # the same C++ pass entered through TensorNeRF.forward + backward() from nmf_amd.trainer.Trainer's loop (fused loss functions):
    # what `extras.module_path` has meant since round 3
    tr_ = Trainer(nerf, params, tape_free=False)
    batches_, f_ = make_batches(nerf, 80, CHUNK, 0, device, distinct=12)
    graph0_ = nerf.operator_graph_forwards
    dt_, rays_, last_, _ = time_train(tr_, batches_, f_, DeviceNoise(device, seed=5), 20, 60, CHUNK, sync)
    if nerf.operator_graph_forwards != graph0_:
        raise SystemExit(f"module_path: {nerf.operator_graph_forwards - graph0_} training forwards left the fused pass for the operator graph")
    out["module_path"] = dict(ms_per_step=1e3 * dt_ / 60, rays_per_s=rays_ / dt_, samples_per_chunk=last_["n_samples"], steps=60,
                              rays_per_step=CHUNK,
                              note="Trainer(tape_free=False): TensorNeRF.forward + backward() through torch.autograd + FusedAdam -- the drop-in "
                                   "operator classes under a training loop that calls forward and backward itself; one ChunkPass node per chunk")
    del tr_
    out["operator_graph"] = reference_loop_ms(20, 6, False)
    out["operator_graph"]["note"] = ("the reference-style loop with nerf.fused_training_pass = False: the autograd operator graph of "
                                     "nmf_amd/functional.py (what rounds 1-4 delivered to that loop; still the path of debug maps / regulariser gradients)")
    return out


def extras(device, params, focal, main_ms=None, main_rays=None):
    """Driver-visible side measurements (N = 1 only, outside the timed region of `value`)."""
    import torch
    from nmf_amd.noise import DeviceNoise
    from nmf_amd.trainer import Trainer
    out = {}

    def sync():
        torch.cuda.synchronize()

    def train_ms(nerf, rays_per_gpu, steps, warmup, chunk=CHUNK, contexts=None):
        tr = Trainer(nerf, params)
        if contexts is not None and tr.fast is not None:
            tr.fast.n_contexts = contexts
        batches, f = make_batches(nerf, steps + warmup, rays_per_gpu, 0, device, distinct=12)
        dt, rays_done, last, _ = time_train(tr, batches, f, DeviceNoise(device, seed=5), warmup, steps, chunk, sync)
        return dict(ms_per_step=1e3 * dt / steps, rays_per_s=rays_done / dt, samples_per_chunk=last["n_samples"],
                    samples_per_chunk_first_step=last["first_n_samples"],
                    steps=steps, rays_per_step=rays_per_gpu)

    nerf, _ = build(device)
    legs_ = None
    try:
        r_ = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", "host_paths"], capture_output=True, text=True, timeout=900)
        lines_ = [ln for ln in r_.stdout.splitlines() if ln.startswith("{")]
        legs_ = json.loads(lines_[-1]) if r_.returncode == 0 and lines_ else None
    except (subprocess.SubprocessError, ValueError):
        legs_ = None
    if legs_ is None:                      # (the fresh process did not finish: the legs in this one, and the line says so)
        legs_ = host_path_legs(device, params)
        for v_ in legs_.values():
            v_["note"] += "  [timed inside the bench process: the fresh-process leg failed]"
    out.update(legs_)
    dt, n, chunk, cold = time_infer(nerf, device, frames=2)
    out["inference"] = dict(rays_per_s=n / dt, s_per_frame=dt / 2, first_frame_s=cold, frame=f"{FRAME}x{FRAME}", chunk=chunk,
                            note="eval mode, render to completion (renderer.py:56-106), rgb/acc outputs, S1 at 128^3; one "
                                 "whole warm-up frame (first_frame_s: the caching allocator grows its pools), two timed")
    # eval_batch_size is a key of the reference's model config (4096 in microfacet_tensorf2.yaml): the evaluation path has no
    # sample budget, so larger chunks only amortise the ~45 dependent launches and four size read-backs of a chunk
    by_chunk = {}
    for c in (16384, 32768):
        dtc, nc, _, _ = time_infer(nerf, device, frames=2, chunk=c)
        by_chunk[str(c)] = nc / dtc
    out["inference"]["rays_per_s_by_eval_batch_size"] = by_chunk
    nerf.model.max_retrace_rays = [1000]
    out["early_phase"] = train_ms(nerf, CHUNK, 40, 10)
    out["early_phase"]["note"] = "max_retrace_rays = 1000 (first 19 chunks after every (re)start, SURVEY F9)"
    nerf.model.max_retrace_rays = [nerf.model.max_brdf_rays[0]]
    out["rays_32768_per_gpu"] = train_ms(nerf, 32768, 8, 4)
    out["rays_32768_per_gpu"]["note"] = ("BASELINE configs[3] per-GPU workload: 8 chunks of 4096 rays under the reference's per-chunk budgets, "
                                         "one optimizer step; the chunks alternate between two chunk contexts (nmf_amd/fast_step.py: the "
                                         "forward of chunk k + 1 next to the backward of chunk k)")
    out["rays_32768_per_gpu_sequential"] = train_ms(nerf, 32768, 8, 4, contexts=1)
    out["rays_32768_per_gpu_sequential"]["note"] = "the same step with ONE chunk context: the chunks strictly one after another (round 5)"
    # the same step with the per-chunk budgets of the reference's config (sampler.max_samples 200 000, model.max_brdf_rays
    # [650 000, 450 000]: sized for a 24 GB card) scaled by 4: two chunks of 16 384 rays, 2.3 GiB peak -- per-ray statistics
    # unchanged (the bounce budget grows with the chunk's weight total), the ~115 dependent launches of a chunk paid twice
    # instead of eight times (tools/big_chunk.py: x2 / x4 / x8)
    scale_budgets(nerf, 4)
    # (five warm-up steps: the allocator's pools for the 4 x larger buffers are still growing in the first ones -- with two, one box of
    #  round 5 timed 10.9 ms here where fresh processes give 6.7-6.8)
    out["rays_32768_per_gpu_budgets_x4"] = train_ms(nerf, 32768, 8, 5, chunk=4 * CHUNK)
    out["rays_32768_per_gpu_budgets_x4"]["note"] = ("the same 32 768-ray optimizer step as 2 chunks of 16 384 rays: sampler.max_samples / "
                                                    "model.max_brdf_rays x 4 (config keys of the reference; `bench.py --budget-scale 4`)")
    del nerf
    torch.cuda.empty_cache()
    per_grid = {}
    for G, _w in SCHEDULE[1:]:
        nerf_g, _ = build(device, grid=G)
        per_grid[G] = train_ms(nerf_g, CHUNK, 30 if G == 300 else 16, 8 if G == 300 else 5)
        del nerf_g
        torch.cuda.empty_cache()
    out["grid_300"] = per_grid[300]
    out["grid_300"]["note"] = "final grid of the schedule: 300^3, 1036 steps per ray, 41 MB of factor tables"
    if main_ms is not None:
        # the reference's 30 000-iteration schedule spends 2000 iterations at 128^3 and 23 000 at 300^3: the
        # rays/s of a whole training run = rays of all iterations / time of all iterations, one 4096-ray chunk per iteration
        rows = {128: dict(ms_per_step=main_ms, rays_per_step=main_rays)}
        rows.update({G: dict(ms_per_step=r["ms_per_step"], rays_per_step=r["rays_per_s"] * r["ms_per_step"] * 1e-3) for G, r in per_grid.items()})
        t_all = sum(w * rows[G]["ms_per_step"] * 1e-3 for G, w in SCHEDULE)
        r_all = sum(w * rows[G]["rays_per_step"] for G, w in SCHEDULE)
        out["schedule_weighted"] = dict(rays_per_s=r_all / t_all, hours_for_30000_iterations=t_all / 3600,
                                        per_grid={str(G): {k: round(v, 3) for k, v in rows[G].items()} for G, _ in SCHEDULE},
                                        note="steady-state step at every grid of the upsampling schedule, weighted by the iterations "
                                             "spent there; kept rays per chunk fall with the grid (200 k-sample budget)")
    out["psnr_at_iter"] = psnr_at_iter(device)
    # bf16 factor tables (BASELINE configs[1]) against fp32 tables, each in a FRESH process = the protocol of the headline
    # number (before the side streams became process-wide, two in-process legs compared the order in which the two trainers
    # were built, not the table type: DESIGN 0.2)
    legs = {}
    for name in ("bf16", "f32"):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--table-dtype", name, "--steps", "80", "--warmup", "20",
                            "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        legs[name] = json.loads(lines[-1]) if r.returncode == 0 and lines else None
    if legs["bf16"] is None or legs["f32"] is None:
        out["bf16_tables"] = dict(error="the fresh-process legs did not finish")
    else:
        out["bf16_tables"] = dict(ms_per_step=legs["bf16"]["ms_per_step"], rays_per_s=legs["bf16"]["value"],
                                  f32_tables_same_protocol_ms=legs["f32"]["ms_per_step"], steps=80, rays_per_step=CHUNK,
                                  samples_per_chunk=legs["bf16"]["config"]["samples_per_chunk"],
                                  note=("BASELINE configs[1]: factor tables read as bf16 by the forward queries (fp32 master copy "
                                        "for Adam and the backward walks, fp32 arithmetic); both numbers are `python bench.py "
                                        "--table-dtype bf16|f32 --steps 80 --warmup 20` in a fresh process each (the workload of "
                                        "the headline number); PSNR delta in docs/DESIGN_rounds_1-5.md"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=60)      # (the first ~50 steps of a process run 3 % slower: allocator growth, clocks)
    ap.add_argument("--rays-per-gpu", type=int, default=CHUNK)
    ap.add_argument("--budget-scale", type=int, default=1,
                    help="per-chunk budgets (sampler.max_samples, model.max_brdf_rays) and the chunk size times this factor")
    ap.add_argument("--grid", type=int, default=GRID)
    ap.add_argument("--mode", choices=("train", "infer"), default="train")
    ap.add_argument("--table-dtype", choices=("f32", "bf16"), default="f32")
    ap.add_argument("--retrace", type=int, default=None,
                    help="max_retrace_rays (default: every secondary ray = the steady state; 1000 = the early phase)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-steps", default=None, help="W,K: W warm-up + K timed CPU steps per phase, no time budget")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--contexts", type=int, default=None, help="chunk contexts of a multi-chunk step (default: NMF_CHUNK_CONTEXTS or 2; 1 = sequential)")
    ap.add_argument("--detail-dir", default=None, help=f"where {DETAIL_FILE} (the full result: per-kernel table, notes, every leg) is written; default: next to bench.py")
    ap.add_argument("--leg", default=None, help="internal: `host_paths` = the host-bound legs of `extras` in this (fresh) process, one JSON line")
    ap.add_argument("--core", action="append", default=[], metavar="ATTR=0|1",
                    help="A/B only: a switch of the C++ pass (csrc/step_core.inc: env_split, overlap ...) set before the warm-up; "
                         "named in config.core_switches")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.leg == "host_paths":
        import torch
        torch.cuda.set_device(0)
        torch.manual_seed(20211200)
        dev_ = torch.device("cuda", 0)
        _, params_ = build(dev_)
        legs = host_path_legs(dev_, params_)
        flush_c_stdio()
        sys.stdout.write(json.dumps(legs) + "\n")
        sys.stdout.flush()
        return

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
    chunk_rays = CHUNK * max(args.budget_scale, 1)
    if args.budget_scale > 1:
        scale_budgets(nerf, args.budget_scale)
    if args.retrace is not None:
        nerf.model.max_retrace_rays = [args.retrace]
    if world > 1:          # every rank trains rank 0's replica; checked (a mismatch raises on every rank)
        from nmf_amd.trainer import broadcast_replica, check_replicas
        broadcast_replica(nerf, src=0)
        check_replicas(nerf, what="after the start-up broadcast")
    timer = RebuildCounter()
    from nmf_amd import hip as hip_mod
    workload = (f"S1 (solid cube, SURVEY 8d), TensoRF {args.grid}^3 (16+24 comps, {args.table_dtype} tables), env 512x1024, "
                f"800x800 camera")

    if args.mode == "infer":
        dt, n_rays, chunk, _cold = time_infer(nerf, device, frames=max(args.steps, 1))
        tt = torch.tensor([dt, float(n_rays)], dtype=torch.float64, device=device)
        if world > 1:
            tmax = tt.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            dt, n_rays = float(tmax[0]), float(tt[1])
        line = None
        if rank == 0:
            line = json.dumps({
                "metric": "inference rays/sec (microfacet_tensorf2, full 800x800 frame, render to completion)",
                "value": n_rays / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": 0,
                "ms_per_step": 1e3 * dt / max(args.steps, 1), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": workload + f", eval mode, {chunk} rays per chunk; stands in for BASELINE configs[4]",
                           "parallelism": f"replicas x{world}" if world > 1 else "dp1"}})
        if world > 1:                         # the JSON line last (see the end of main)
            flush_c_stdio()
            dist.barrier()
            dist.destroy_process_group()
        flush_c_stdio()
        if line is not None:
            sys.stdout.write(line + "\n")
            sys.stdout.flush()
        return

    trainer = Trainer(nerf, params, world_size=world, rank=rank)
    if args.contexts is not None and trainer.fast is not None:
        trainer.fast.n_contexts = max(1, args.contexts)
    for kv in args.core:
        k, v = kv.split("=")
        trainer.fast.set_switch(k, bool(int(v)))
    noise = DeviceNoise(device, seed=1000 + rank)
    batches, focal = make_batches(nerf, args.warmup + args.steps, args.rays_per_gpu, rank, device)
    fx = hip_mod.HOST_EXT
    timed_calls = fx is not None and hasattr(fx, "kernel_timing_begin") and trainer.fast is not None and trainer.fast.core() is not None
    n_probe = min(args.warmup, 8) if timed_calls else 0
    timer.enabled = False
    dominant = None
    # Process start-up, not warm-up: the first ~50 optimizer steps of a process grow the caching allocator's pools and ramp the clocks
    # (3-5 % slower; the driver's `--warmup 5` landed 5 % above the 60-warm-up figure in round 4).  They are run here, untimed, before
    # the W warm-up steps the command line asks for -- the K timed steps are the steady state the metric names.  Named in the line
    # (config.startup_steps).
    startup_steps = max(0, STARTUP_STEPS - args.warmup)
    for i in range(startup_steps):
        trainer.step(*batches[i % len(batches)], focal, noise=noise, update_controllers=False, fixed_chunk=chunk_rays)
    for i in range(args.warmup):
        if n_probe and i == args.warmup - n_probe:
            fx.kernel_timing_begin()                # the last warm-up steps find the dominant KERNEL of this workload
        trainer.step(*batches[i % len(batches)], focal, noise=noise, update_controllers=False, fixed_chunk=chunk_rays)
    if n_probe:
        probe = {k: v for k, v in fx.kernel_timing_end().items() if not k.startswith("@")}
        # (among the kernels kernel_models() prices: a latency-bound kernel without a work model has no fraction to report)
        priced = {k for k in kernel_models(dict.fromkeys(("B", "M0", "M1", "R0", "R1", "Mb0", "Mb1", "N"), 1), 1) if not k.startswith("group:")}
        dominant = max(({k: v for k, v in probe.items() if k in priced} or probe).items(), key=lambda kv: kv[1][0])[0]
    sync()
    timer.enabled = True
    if dominant:
        fx.kernel_timing_begin(dominant)            # inside the timed region: events around the launches of the dominant kernel only
    dt, rays_done, last, comm_ms = time_train(trainer, batches, focal, noise, 0, args.steps, chunk_rays, sync)
    dom_live = fx.kernel_timing_end().get(dominant) if dominant else None
    timer.enabled = False

    tt = torch.tensor([dt, float(rays_done), 1.0], dtype=torch.float64, device=device)
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt_max, rays_all, ranks_seen = float(tmax[0]), float(tt[1]), int(round(float(tt[2])))
    else:
        dt_max, rays_all, ranks_seen = dt, float(rays_done), 1

    chunks_per_step = -(-args.rays_per_gpu // chunk_rays)
    if min(timer.rebuilds.values()) < args.steps:
        raise SystemExit(f"derived tables were not rebuilt every step: {timer.rebuilds} for {args.steps} steps")
    # ---- after the timed region: every kernel launch of the step timed (events on the launching stream), 30 more steps
    table, sizes, n_params = None, None, sum(p.numel() for p in nerf.parameters() if p.requires_grad)
    main_stream_us = None
    if timed_calls:                       # (every rank: the steps contain the gradient all-reduce)
        n_inst = 30
        main_id = torch.cuda.current_stream().cuda_stream
        sync()
        t_inst = time.perf_counter()
        fx.kernel_timing_begin()
        for i in range(n_inst):
            trainer.step(*batches[i % len(batches)], focal, noise=noise, update_controllers=False, fixed_chunk=chunk_rays)
        timing = fx.kernel_timing_end()                # (waits for the recorded events)
        wall_inst_us = 1e6 * (time.perf_counter() - t_inst) / n_inst
        if f"@{main_id}" in timing:
            main_stream_us = 1e3 * timing[f"@{main_id}"][0] / n_inst
        ls = trainer.fast.last_sizes
        sizes = dict(B=int(ls["rays"]), M0=int(ls["n_samples"][0]), M1=int(ls["n_samples"][1]) if len(ls["n_samples"]) > 1 else 0,
                     R0=int(ls["n_rays"][0]), R1=int(ls["n_rays"][1]) if len(ls["n_rays"]) > 1 else 0,
                     Mb0=int(ls["n_rows"][0]), Mb1=int(ls["n_rows"][1]) if len(ls["n_rows"]) > 1 else 0, N=int(nerf.sampler.nSamples))
        # (sizes are those of ONE chunk; the table is per chunk when a step has several)
        table = per_kernel_table(timing, n_inst * chunks_per_step, sizes, n_params, counters_summary()) if rank == 0 else None
    if rank == 0:
        ctr = counters_summary()
        ctr_meta = {k: ctr.get(k) for k in ("tag", "commit", "command")} if ctr else None
        roof = {"note": "the host extension with kernel timing is not available: no per-kernel table"}
        if table:
            dname = dominant if dominant in table else next(k_ for k_ in table if not k_.startswith("group:"))
            drow = table[dname]
            models = kernel_models(sizes, n_params)
            bound, work, peak, unit = models.get(dname, ("latency", None, None, None))
            live_us = 1e3 * dom_live[0] / max(dom_live[1], 1) if dom_live else None          # per launch, inside the timed region
            launches_per_chunk = dom_live[1] / (args.steps * chunks_per_step) if dom_live else drow["launches_per_step"]
            # achieved = useful work of one launch / that launch's duration (launches of different sizes: work and time are both
            # summed over the launches of a chunk, i.e. work per chunk / (launches per chunk x average launch duration))
            achieved = work / (launches_per_chunk * live_us * 1e-6) if (work and live_us) else None
            scale = {"TFLOP/s": 1e12, "GB/s": 1e9, "G updates/s": 1e9, "T lane-ops/s": 1e12}.get(unit, 1.0)
            survey_b, needed_b = step_bytes(sizes, args.grid, n_params)
            ms_step = 1e3 * dt_max / args.steps
            ck = (ctr or {}).get("kernels", {}).get(counter_key(dname), {}) if ctr else {}
            extra = issued_beside(dname, sizes)
            dev_sum = round(sum(r["us_per_step"] for k_, r in table.items() if not k_.startswith("group:")), 1)
            roof = {
                # the dominant KERNEL of the step by summed device time, found in the warm-up and timed with HIP events on its
                # launching stream INSIDE the timed region; frac = useful work / duration / peak (kernel_models())
                "kernel": dname, "bound": "mfma" if bound == "mfma" else ("hbm" if bound in ("hbm", "atomics") else bound),
                "bound_detail": {"mfma": "useful fp32 FLOP on the matrix cores against the 157.3 TFLOP/s fp32-input MFMA peak (= the fp32 vector peak)",
                                 "valu": "useful fp32 FLOP / lane operations against the VALU peak", "hbm": "compulsory bytes against 8 TB/s",
                                 "atomics": "table updates against the rate of the memory-side atomic units"}.get(bound, bound),
                "achieved": achieved / scale if achieved else None, "peak": peak / scale if peak else None,
                "unit": unit, "frac": (achieved / peak) if (achieved and peak) else None,
                "issued_bf16_frac": (extra["issued_bf16_flop"] / (launches_per_chunk * live_us * 1e-6) / extra["issued_peak"]) if (extra and live_us) else None,
                "traffic": ck.get("hbm_bytes_per_launch"),
                "counters_of_kernel": {k: ck.get("derived", {}).get(k) for k in ("alu_busy", "mfma_busy", "waves_per_simd", "hbm_frac", "l2_frac")} if ck else None,
                "launches": dom_live[1] if dom_live else None, "avg_launch_us": live_us,
                "useful_work_per_chunk": work,
                "work_model": "kernel_models() in bench.py / DESIGN.md section 4: SURVEY 8(d)'s FLOP per unit x the units of `sizes_per_step`",
                "per_kernel": table, "sizes_per_step": sizes,
                "step": {"wall_us": round(1e3 * ms_step, 1), "main_stream_kernel_us": round(main_stream_us, 1) if main_stream_us else None,
                         # the main stream carries the dependency chain of the step (side streams only ever run next to it): the share
                         # of the wall time in which a kernel of the chain is executing
                         # (both from the instrumented steps: an event pair around every launch stretches them)
                         "wall_us_instrumented": round(wall_inst_us, 1),
                         "critical_path_frac": round(min(main_stream_us / wall_inst_us, 1.0), 4) if main_stream_us else None,
                         "device_time_sum_us": dev_sum,
                         "survey_8d_bytes": survey_b, "survey_8d_over_hbm": survey_b / (ms_step * 1e-3) / (HBM_PEAK_GBS * 1e9),
                         "needed_bytes": needed_b, "needed_over_l2": needed_b / (ms_step * 1e-3) / (L2_PEAK_GBS * 1e9),
                         "note": "survey_8d_over_hbm is SURVEY 8(d)'s byte model kept for the record: the tables (7.5 MB at 128^3) are L2 / MALL "
                                 "resident and it exceeds 1, so it bounds nothing.  The step is a dependent chain of ~75 launches on the main "
                                 "stream with side streams next to it in the backward (DESIGN 0)"},
                "counters": ctr_meta,
            }
        out = {
            "metric": f"train rays/sec (microfacet_tensorf2, {chunk_rays}-ray chunks, steady state)",
            "value": rays_all / dt_max, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.table_dtype == "f32" else "bf16 tables / f32 arithmetic",
            "data": "synthetic",
            # (<= 300 characters: compact_line keeps it whole; the a20 note -- in the steady state the score + sort of
            #  models/microfacet.py:475-537 selects every secondary ray and is skipped -- is `config.retrace` in the detail file)
            "config": {"workload": f"BASELINE configs[{1 if args.rays_per_gpu == CHUNK else 3}] on scene " + workload
                                   + f", {args.rays_per_gpu} rays/GPU/step in {chunks_per_step} chunk(s) of {chunk_rays}"
                                   + (f" (per-chunk budgets x{args.budget_scale})" if args.budget_scale > 1 else "")
                                   + ", fwd+bwd+all-reduce+Adam, "
                                   + ("steady state (all secondary rays re-traced)" if args.retrace is None
                                      else f"{args.retrace} secondary rays re-traced"),
                       "retrace": ("all secondary rays re-traced: the a20 score + sort of models/microfacet.py:475-537 selects every ray in "
                                   "the steady state and is skipped" if args.retrace is None else f"{args.retrace} secondary rays re-traced")
                                  + "; lego / ship are not available offline: scene S1 of SURVEY 8(d)",
                       "rays_per_gpu": args.rays_per_gpu, "chunks_per_step": chunks_per_step, "grid": args.grid,
                       "chunks_in_flight": min(chunks_per_step, trainer.fast.n_contexts) if trainer.fast is not None else 1,
                       "operator_graph_fallbacks": int(nerf.operator_graph_forwards),
                       "samples_per_chunk": last["n_samples"], "samples_per_chunk_first_step": last["first_n_samples"],
                       "table_rebuilds_in_timed_region": dict(timer.rebuilds),
                       "parallelism": f"dp{world}", "ranks_seen": ranks_seen,
                       "backend": backend if (world > 1 or single_rank_comm) else None,
                       "comm_ms_per_step": comm_ms, "comm_bytes_per_step": last["comm_bytes"],
                       # time the main stream waited for the step's collectives: what is left of the early bucket (BRDF MLP, heads,
                       # environment map: started inside the last chunk's backward, next to the field walks) + the late bucket
                       "comm_exposed_ms": (last["comm_exposed_ms"] if last["comm_bytes"] else None),
                       "host_cpu_ms_per_step": last.get("host_cpu_ms_per_step"),
                       "startup_steps": startup_steps,
                       "host_pass": "C++ (csrc/step_core.inc)" if (trainer.fast is not None and trainer.fast.core() is not None) else "python",
                       **({"core_switches": args.core} if args.core else {})},
            "roofline": roof,
        }
        if world == 1 and not args.no_extras and args.rays_per_gpu == CHUNK and args.grid == GRID and args.budget_scale == 1 \
                and args.table_dtype == "f32":
            del trainer
            # the legs below include host-bound loops (the reference-style loop, the module path): the step batches and the tables of
            # the run above are released and what stays alive is taken out of the garbage collector's generations, so that its
            # full collections do not walk the remains of the main run inside those loops
            import gc
            batches = nerf = noise = table = timing = None
            gc.collect()
            torch.cuda.empty_cache()
            gc.freeze()
            out["extras"] = extras(device, params, focal, main_ms=1e3 * dt_max / args.steps, main_rays=rays_all / args.steps)
            out["psnr_at_iter"] = out["extras"].get("psnr_at_iter")
        if world == 1 and not args.no_cpu_baseline:
            if args.cpu_baseline_steps:
                w_, k_ = (int(v) for v in args.cpu_baseline_steps.split(","))
                out["cpu_baseline"] = cpu_baseline(budget_s=None, warm=w_, timed=k_)
            else:
                out["cpu_baseline"] = cpu_baseline()
        detail = os.path.join(args.detail_dir or ROOT, DETAIL_FILE)
        try:
            with open(detail, "w") as f:
                json.dump(out, f, indent=1)
            shown = os.path.relpath(detail, ROOT) if detail.startswith(ROOT) else detail
        except OSError as e:                       # (a read-only checkout: the line still goes out)
            shown = f"not written ({e.__class__.__name__})"
        line = compact_line(out, shown)
    else:
        line = None
    # The JSON line is the LAST thing this job writes to stdout: RCCL prints a version banner through C stdio (buffered when
    # stdout is a pipe or a file, so it would otherwise come out at process exit, BEHIND the line), and the other ranks
    # share rank 0's stdout under torch.distributed.run.  Every rank tears the process group down and flushes its C
    # stdio first; rank 0 prints after the last barrier.
    if world > 1 or single_rank_comm:
        flush_c_stdio()
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if line is not None:
        if world > 1:
            time.sleep(0.3)                   # the other ranks flush what their teardown printed
        sys.stdout.write(line + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
