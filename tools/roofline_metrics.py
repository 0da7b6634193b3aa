"""Per-kernel counter figures of the steady-state step from the rocprofv3 runs of tools/profile_round.sh (one counter group per
run, `--pmc <group> --kernel-trace`).
    python tools/roofline_metrics.py --collect <tag> <out dir> <repo root> <pmc dir prefix> <kernel-stats dir>   (profile_round.sh)
    python tools/roofline_metrics.py gpurun_out/<tag>_roofline.json          re-derive an existing file

R4: every figure of a kernel comes from ONE kind of run.  Counter runs serialise the kernels, so their
launch durations differ from the `--kernel-trace --stats` run (where up to seven streams overlap), and the first launches of a
bench run have other sizes than the steady state; rounds 1-3 divided cycles of the counter runs by durations of the stats run,
averaged over all launches, and got "clocks" of 1.4-4.5 GHz.  Now: per counter run, only the launches of the LAST optimizer steps
(between the last k_adam dispatches) are used, durations are taken from the dispatch timestamps of the SAME counter run, and a
kernel whose (cycles - dispatch cycles) / duration is outside 1.8-2.5 GHz (the part runs at <= 2.4 GHz) gets no `derived` block;
GRBM_GUI_ACTIVE also counts the dispatch of a launch (~18 k cycles, calibrated per file from its shortest kernels and subtracted).  A rejected kernel of >= 20 us fails the tool.

Normalisation on MI355X (8 XCDs x 4 SEs, 256 CUs x 4 SIMDs; checked on k_vm_bwd_brick: 16.15 M v_mfma_f32_16x16x4_f32 x 32
cycles/SIMD = 516.8 M = SQ_VALU_MFMA_BUSY_CYCLES exactly):
  GRBM_GUI_ACTIVE            summed over the 8 XCDs          -> active cycles of the launch = GRBM_GUI_ACTIVE / 8 - dispatch cycles
  SQ_VALU_MFMA_BUSY_CYCLES   cycles, summed over 1024 SIMDs  -> mfma_busy = it / (1024 * active cycles)
  SQ_ACTIVE_INST_VALU        quad-cycles (4 clocks) of VALU ISSUE, summed over SIMDs; an MFMA counts as one quad-cycle
                             here whatever its length (k_brdf_mlp_bwd: 16.7 M for 16.6 M VALU instructions of which
                             2.7 M are 64-cycle MFMAs)       -> valu_issue_busy = 4 * it / (1024 * active cycles)
  alu_busy                   = mfma_busy + valu_issue_busy - 4 * SQ_INSTS_MFMA / (1024 * active cycles).  The two ADD UP on
                             gfx950: while an fp32 MFMA executes, no other VALU instruction of any wave issues on that SIMD
                             (tools/ub/coexec.hip: an MFMA wave and an FMA wave on one SIMD take longer than one after the
                             other; SQ_VALU_MFMA_COEXEC_CYCLES = 0 in every kernel here) -- the fp32 matrix rate equals the
                             packed-FMA rate, so the ceiling of an fp32-MFMA kernel is alu_busy = 1, not mfma_busy = 1.
  SQ_WAVE_CYCLES             quad-cycles of resident waves   -> occupancy = 4 * it / (1024 * active cycles) waves per SIMD
  TCC_REQ_sum                128-byte L2 requests            -> l2_frac against 34.5 TB/s
  FETCH_SIZE / WRITE_SIZE    KB at the fabric (uncorrected, see MI355X_MICROARCH.md "HBM") -> hbm_frac against 8 TB/s
"""
import json
import sys

HBM, L2, SIMDS = 8.0e12, 34.5e12, 1024


DISPATCH_CYCLES = [0.0]      # GRBM_GUI_ACTIVE / 8 of an (almost) empty launch: calibrated per file from its shortest kernels


def calibrate_dispatch_cycles(kernels):
    """GRBM_GUI_ACTIVE also counts the dispatch of the launch itself: r04_a, k_bins_partial 4.8 us -> 29.6 k cycles, k_segment_sum_wide
    5.7 us -> 32.0 k: ~18 k cycles (7.5 us at 2.4 GHz) on top of duration x clock for every kernel, which made 20-30 us launches read as
    2.6-3.7 GHz.  Estimated as the median of (GRBM / 8 - 2400 cycles/us x duration) over the launches shorter than 8 us."""
    est = []
    for v in kernels.values():
        c, t = v.get("counters_per_launch", {}), v.get("avg_launch_us")
        if t and t < 8.0 and c.get("GRBM_GUI_ACTIVE"):
            est.append(c["GRBM_GUI_ACTIVE"] / 8 - 2400.0 * t)
    est.sort()
    return max(est[len(est) // 2], 0.0) if est else 0.0


def derive(name, rec):
    c = rec.get("counters_per_launch", {})
    t = rec.get("avg_launch_us")
    if not c or not t:
        return rec
    t *= 1e-6
    d = {}
    act = c.get("GRBM_GUI_ACTIVE")
    cyc = max(act / 8 - DISPATCH_CYCLES[0], 1.0) if act else None
    if cyc:
        d["active_cycles"] = round(cyc)
        d["clock_GHz"] = round(cyc / t / 1e9, 2)
        if c.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            d["mfma_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * cyc), 4)
        if c.get("SQ_ACTIVE_INST_VALU") is not None:
            d["valu_issue_busy"] = round(4 * c["SQ_ACTIVE_INST_VALU"] / (SIMDS * cyc), 4)
            d["alu_busy"] = round(d.get("mfma_busy", 0.0) + d["valu_issue_busy"] - 4 * c.get("SQ_INSTS_MFMA", 0) / (SIMDS * cyc), 4)
        if c.get("SQ_WAVE_CYCLES") is not None:
            d["waves_per_simd"] = round(4 * c["SQ_WAVE_CYCLES"] / (SIMDS * cyc), 2)
        if c.get("SQ_WAIT_INST_ANY") is not None and c.get("SQ_WAVE_CYCLES"):
            d["issue_stall_frac_of_wave_cycles"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)
        if c.get("SQ_WAIT_ANY") is not None and c.get("SQ_WAVE_CYCLES"):
            d["waitcnt_frac_of_wave_cycles"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
    if c.get("FETCH_SIZE") is not None and c.get("WRITE_SIZE") is not None:
        b = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        d["hbm_bytes"] = round(b)
        d["hbm_frac"] = round(b / t / HBM, 4)
    if c.get("TCC_REQ_sum") is not None:
        d["l2_frac"] = round(c["TCC_REQ_sum"] * 128 / t / L2, 4)
        if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None:
            d["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1), 4)
    if c.get("SQ_INSTS_MFMA"):
        flop = c["SQ_INSTS_MFMA"] * (4096 if "brdf_mlp" in name else 2048)     # 32x32x2 / 16x16x4 f32-input MFMA
        d["mfma_tflops"] = round(flop / t / 1e12, 2)
        d["mfma_frac_of_157.3_tflops"] = round(flop / t / 157.3e12, 4)
    fr = {k: d[k] for k in ("alu_busy", "hbm_frac", "l2_frac") if k in d}
    if fr:
        d["bound"] = max(fr, key=fr.get)
        d["frac"] = fr[d["bound"]]
    out = {k: v for k, v in rec.items() if k in ("avg_launch_us", "launches_profiled", "counters_per_launch", "note",
                                                 "hbm_bytes_per_launch", "avg_launch_us_by_run",
                                                 "avg_launch_us_overlapped_all_launches")}
    out["derived"] = d
    if "hbm_bytes" in d:
        out["hbm_bytes_per_launch"] = d["hbm_bytes"]
    return out


CLOCK_LO, CLOCK_HI = 1.8, 2.5

# kernels of the steady-state step, by substring of the demangled name -> report key
KEYS = {"k_vm_bwd_density<false": "k_vm_bwd_density<value>", "k_vm_bwd_density<true": "k_vm_bwd_density<normal>",
        "k_vm_bwd_brick<true": "k_vm_bwd_brick<density>", "k_vm_sigma": "k_vm_sigma", "k_vm_rows_dn": "k_vm_rows_dn",
        "k_vm_app_rows": "k_vm_app_rows", "k_vm_bwd_brick<false": "k_vm_bwd_brick<appearance>",
        "k_brdf_mlp_bwd": "k_brdf_mlp_bwd", "k_brdf_mlp_fwd": "k_brdf_mlp_fwd", "k_brdf_mlp_reduce": "k_brdf_mlp_reduce",
        "k_env_lookup_bwd": "k_env_lookup_bwd", "k_env_lookup_fwd": "k_env_lookup_fwd", "k_vm_fwd": "k_vm_fwd",
        "k_march_count16": "k_march_count16", "k_march_fill16": "k_march_fill16", "k_brick_records": "k_brick_records",
        "k_plan_hist": "k_plan_hist", "k_plan_place": "k_plan_place", "k_ggx_rays_bwd": "k_ggx_rays_bwd",
        "k_composite_bwd": "k_composite_bwd", "k_adam": "k_adam", "k_env_bin_count": "k_env_bin_count",
        "k_env_bin_scatter": "k_env_bin_scatter", "k_env_bin_accum": "k_env_bin_accum", "k_segment_sum_wide": "k_segment_sum_wide",
        "k_march_count(": "k_march_count", "k_bins_final": "k_bins_final", "k_bins_partial": "k_bins_partial"}


def key_of(name):
    for sub, k in KEYS.items():
        if sub in name:
            return k
    return None


def collect(tag, out, root, pmc_prefix, ks_dir, last_steps=10):
    import collections
    import csv
    import glob
    import os
    import subprocess
    per = collections.defaultdict(lambda: collections.defaultdict(list))      # key -> counter -> [values of steady-state launches]
    dur = collections.defaultdict(lambda: collections.defaultdict(list))      # key -> counter run -> [us of the same launches]
    for d in sorted(glob.glob(pmc_prefix + "*")):
        fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not fs:
            continue
        rows = list(csv.DictReader(open(fs[0])))
        if not rows:
            continue
        did = "Dispatch_Id" if "Dispatch_Id" in rows[0] else "Correlation_Id"
        adam = sorted({int(r[did]) for r in rows if "k_adam" in r["Kernel_Name"]})
        lo = adam[-1 - last_steps] if len(adam) > last_steps else (adam[0] if adam else -1)
        hi = adam[-1] if adam else 1 << 62
        seen = set()
        for r in rows:
            i = int(r[did])
            if not (lo < i <= hi):
                continue
            k = key_of(r["Kernel_Name"])
            if k is None:
                continue
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if (i, k) not in seen and r.get("Start_Timestamp") and r.get("End_Timestamp"):
                seen.add((i, k))
                dur[k][os.path.basename(d)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    # overlapped durations of the --kernel-trace --stats run, for reference only
    stats = {}
    for f in glob.glob(ks_dir + "/**/*kernel_stats.csv", recursive=True)[:1]:
        for r in csv.DictReader(open(f)):
            k = key_of(r["Name"])
            if k:
                a = stats.setdefault(k, [0, 0.0])
                a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
    kernels = {}
    for k, ctrs in per.items():
        rec = {"counters_per_launch": {c: round(sum(v) / len(v), 1) for c, v in sorted(ctrs.items())},
               "launches_profiled": max(len(v) for v in ctrs.values())}
        runs = dur.get(k, {})
        if runs:
            allv = [x for v in runs.values() for x in v]
            rec["avg_launch_us"] = round(sum(allv) / len(allv), 2)          # serialised (counter runs), steady-state launches only
            rec["avg_launch_us_by_run"] = {r: round(sum(v) / len(v), 2) for r, v in sorted(runs.items())}
        if k in stats and stats[k][0]:
            rec["avg_launch_us_overlapped_all_launches"] = round(stats[k][1] / stats[k][0] / 1e3, 2)
        kernels[k] = rec
    try:
        commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        try:
            commit = open(root + "/.git_sha").read().strip()
        except OSError:
            commit = "unknown"
    doc = {"tag": tag, "commit": commit, "bench_args": os.environ.get("BENCH_ARGS", ""),
           "command": "rocprofv3 --pmc <group> --kernel-trace -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras "
                      "[bench_args], one run per counter group (tools/profile_round.sh); counters AND durations of a kernel from the "
                      f"launches of the last {last_steps} optimizer steps of those runs (kernels serialised by the counter collection)",
           "units": "FETCH_SIZE / WRITE_SIZE in KB as reported (uncorrected: MI355X_MICROARCH.md calibrates the x2 only for 16 B/lane "
                    "streaming reads, this path gathers 64-192 B runs); *_frac relative to 8 TB/s HBM, 34.5 TB/s L2, 157.3 TFLOP/s f32 MFMA",
           "kernels": kernels}
    path = f"{out}/{tag}_roofline.json"
    json.dump(doc, open(path, "w"), indent=1)
    return path


def main(path):
    doc = json.load(open(path))
    DISPATCH_CYCLES[0] = calibrate_dispatch_cycles(doc["kernels"])
    doc["dispatch_cycles_subtracted"] = round(DISPATCH_CYCLES[0])
    doc["kernels"] = {k: derive(k, v) for k, v in doc["kernels"].items()}
    bad = []
    for k, v in doc["kernels"].items():
        d = v.get("derived")
        if d and "clock_GHz" in d and not (CLOCK_LO <= d["clock_GHz"] <= CLOCK_HI):
            v["rejected"] = (f"cycles / duration = {d['clock_GHz']} GHz is outside {CLOCK_LO}-{CLOCK_HI}: the counters and the duration "
                             "do not describe the same launches (or the launch is too short for GRBM_GUI_ACTIVE)")
            v["derived_rejected"] = v.pop("derived")
            if v.get("avg_launch_us", 0) >= 20:
                bad.append(k)
    doc["normalisation"] = __doc__.split("Normalisation", 1)[1].strip()
    json.dump(doc, open(path, "w"), indent=1)
    for k, v in doc["kernels"].items():
        if "derived" in v:
            print(f"{k:30s} {v['avg_launch_us']:8.1f} us  " + "  ".join(f"{a}={b}" for a, b in v["derived"].items()
                                                                            if a not in ("active_cycles", "hbm_bytes")))
        elif "rejected" in v:
            print(f"{k:30s} {v.get('avg_launch_us', 0):8.1f} us  REJECTED clock {v['derived_rejected'].get('clock_GHz')} GHz")
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "--collect":
        target = collect(*sys.argv[2:7])
    else:
        target = sys.argv[1]
    rejected = main(target)
    if rejected:
        print("REJECTED (clock out of range on a launch of >= 20 us):", rejected)
        sys.exit(1)
