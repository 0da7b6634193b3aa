#!/bin/bash
# Collect the per-round rocprofv3 evidence on an MI355X box:  tools/profile_round.sh <tag>   (e.g. r01_q)
# Writes gpurun_out/<tag>_bench.json, <tag>_bench_kernel_stats.csv, <tag>_pmc_{FETCH,WRITE}_SIZE_per_kernel.csv and
# pmc_vm_bwd.json; copy what is to be judged into profiles/.  PMC passes are separate runs with --kernel-trace only.
# Optional: BENCH_ARGS="--grid 300" (or "--retrace 1000", "--mode infer --steps 2") profiles another regime: the first, full
# bench.py run then also takes those arguments (without extras / CPU baseline).
set -u
TAG=${1:-rXX}
BENCH_ARGS=${BENCH_ARGS:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
if [ -n "$BENCH_ARGS" ]; then
  python "$ROOT/bench.py" $BENCH_ARGS --no-extras --no-cpu-baseline > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
else
  python "$ROOT/bench.py" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
fi
rm -rf /tmp/prof_ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extras $BENCH_ARGS > "$OUT/${TAG}_prof_bench.log" 2>&1
cp "$(find /tmp/prof_ks -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_bench_kernel_stats.csv"
# steady-state view: only the kernels between the last optimizer launches (setup / warm-up launches excluded)
python - "$(find /tmp/prof_ks -name '*kernel_trace.csv' | head -1)" "$OUT/${TAG}_steady_state_per_step.csv" <<'PY'
import csv, sys, collections, re
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
adam = [i for i, r in enumerate(rows) if "k_adam" in r[2]]
n = min(4, len(adam) - 1)
lo, hi = adam[-1 - n] + 1, adam[-1] + 1
agg = collections.defaultdict(lambda: [0, 0])
busy = 0
for s, e, k in rows[lo:hi]:
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"\(.*", "", k)[:90]
    agg[k][0] += 1
    agg[k][1] += e - s
    busy += e - s
span = rows[hi - 1][1] - rows[lo][0]
with open(sys.argv[2], "w") as o:
    o.write(f"# {n} steady-state steps: wall span {span / n / 1e3:.1f} us/step, kernel time {busy / n / 1e3:.1f} us/step, {sum(v[0] for v in agg.values()) / n:.0f} launches/step\n")
    o.write("kernel,launches_per_step,us_per_step,pct_of_kernel_time\n")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        o.write(f"\"{k}\",{c / n:.2f},{t / n / 1e3:.1f},{100 * t / busy:.1f}\n")
print(open(sys.argv[2]).read()[:6000])
PY
# ---- counters, one group per run (rocprofv3 --pmc with --kernel-trace only; 8 SQ slots / 4 TCC slots per pass,
#      FETCH_SIZE costs 3 TCC slots and WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots")
PMC_SETS="FETCH_SIZE WRITE_SIZE SQ_WAVES,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,SQ_INSTS_VALU,SQ_INSTS_MFMA,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_ANY,SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_INST_CYCLES_VMEM,SQ_ACTIVE_INST_LDS,SQ_INSTS_VMEM_WR TCC_HIT_sum,TCC_MISS_sum,TCC_REQ_sum TCC_EA0_RDREQ_sum,TCC_EA0_WRREQ_sum,TCC_EA0_ATOMIC_sum GRBM_GUI_ACTIVE,GRBM_COUNT"
i=0
for G in $PMC_SETS; do
  i=$((i+1))
  rm -rf /tmp/prof_pmc_$i
  timeout 300 rocprofv3 --pmc $(echo $G | tr ',' ' ') --kernel-trace --output-format csv -d /tmp/prof_pmc_$i -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-extras $BENCH_ARGS > "$OUT/${TAG}_pmc_$i.log" 2>&1 || echo "counter group $G failed (see ${TAG}_pmc_$i.log)"
done
python - "$TAG" "$OUT" "$ROOT" <<'PY'
import csv, glob, json, os, subprocess, sys, collections, re
tag, out, root = sys.argv[1], sys.argv[2], sys.argv[3]
# kernels of the steady-state step, by substring of the demangled name -> report key
KEYS = {"k_vm_bwd_density<false": "k_vm_bwd_density<value>", "k_vm_bwd_density<true": "k_vm_bwd_density<normal>",
        "k_vm_bwd_brick<true": "k_vm_bwd_brick<density>", "k_vm_sigma": "k_vm_sigma", "k_vm_rows_dn": "k_vm_rows_dn",
        "k_vm_app_rows": "k_vm_app_rows",
        "k_vm_bwd_brick<false": "k_vm_bwd_brick<appearance>",
        "k_brdf_mlp_bwd": "k_brdf_mlp_bwd", "k_brdf_mlp_fwd": "k_brdf_mlp_fwd", "k_brdf_mlp_reduce": "k_brdf_mlp_reduce", "k_env_lookup_bwd": "k_env_lookup_bwd",
        "k_env_lookup_fwd": "k_env_lookup_fwd", "k_vm_fwd": "k_vm_fwd", "k_march_count16": "k_march_count16",
        "k_march_fill16": "k_march_fill16", "k_brick_scatter": "k_brick_scatter", "k_ggx_rays_bwd": "k_ggx_rays_bwd",
        "k_composite_bwd": "k_composite_bwd", "k_adam": "k_adam", "k_env_bin_count": "k_env_bin_count",
        "k_env_bin_scatter": "k_env_bin_scatter", "k_env_bin_accum": "k_env_bin_accum", "k_segment_sum_wide": "k_segment_sum_wide",
        "k_march_count(": "k_march_count", "k_brick_hist": "k_brick_hist", "k_bins_final": "k_bins_final"}
def key_of(name):
    for sub, k in KEYS.items():
        if sub in name:
            return k
    return None
sums = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sorted(glob.glob("/tmp/prof_pmc_*")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        k = key_of(r["Kernel_Name"])
        if k is None:
            continue
        c = r["Counter_Name"]
        sums[k][c] += float(r["Counter_Value"])
        launches[k][c] += 1
# durations from the --stats run of the same command
dur = {}
ks = glob.glob("/tmp/prof_ks/**/*kernel_stats.csv", recursive=True)
if ks:
    for r in csv.DictReader(open(ks[0])):
        k = key_of(r["Name"])
        if k:
            d = dur.setdefault(k, [0, 0.0])
            d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"])
HBM, L2, CLK, CUS = 8.0e12, 34.5e12, 2.4e9, 256
kernels = {}
for k in sums:
    per = {c: sums[k][c] / max(launches[k][c], 1) for c in sums[k]}
    rec = {"counters_per_launch": {c: round(v, 1) for c, v in sorted(per.items())}}
    if k in dur and dur[k][0]:
        t = dur[k][1] / dur[k][0] * 1e-9
        rec["avg_launch_us"] = round(t * 1e6, 2)
        rec["launches_profiled"] = dur[k][0]
        fetch, write = per.get("FETCH_SIZE"), per.get("WRITE_SIZE")
        if fetch is not None and write is not None:
            rec["hbm_bytes_per_launch"] = (fetch + write) * 1024
            rec["hbm_frac"] = round((fetch + write) * 1024 / t / HBM, 4)
        if "TCC_REQ_sum" in per:
            rec["l2_bytes_per_launch"] = per["TCC_REQ_sum"] * 128          # 128-byte L2 lines
            rec["l2_frac"] = round(per["TCC_REQ_sum"] * 128 / t / L2, 4)
        if per.get("TCC_HIT_sum") is not None and per.get("TCC_MISS_sum") is not None:
            rec["l2_hit_rate"] = round(per["TCC_HIT_sum"] / max(per["TCC_HIT_sum"] + per["TCC_MISS_sum"], 1), 4)
        busy = per.get("SQ_BUSY_CYCLES")
        if per.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and busy:
            rec["mfma_busy"] = round(per["SQ_VALU_MFMA_BUSY_CYCLES"] / busy, 4)
        if per.get("SQ_ACTIVE_INST_VALU") is not None and per.get("SQ_WAVE_CYCLES"):
            rec["valu_issue_frac_of_wave_cycles"] = round(per["SQ_ACTIVE_INST_VALU"] / per["SQ_WAVE_CYCLES"], 4)
        if per.get("SQ_INSTS_MFMA") is not None:
            rec["mfma_insts_per_launch"] = per["SQ_INSTS_MFMA"]
            # v_mfma_f32_16x16x4_f32: 2048 FLOP; v_mfma_f32_32x32x2_f32: 4096 FLOP (brdf MLP kernels)
            flop = per["SQ_INSTS_MFMA"] * (4096 if "brdf_mlp" in k else 2048)
            rec["mfma_tflops"] = round(flop / t / 1e12, 2)
            rec["mfma_frac_of_157.3"] = round(flop / t / 157.3e12, 4)
    kernels[k] = rec
walk = kernels.get("k_vm_bwd_density<value>", kernels.get("k_vm_bwd_density<normal>", kernels.get("k_vm_bwd_brick<density>", {})))
try:
    commit = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
except Exception:
    # no .git on the GPU box: the container writes the SHA next to the sources before the snapshot is taken (tools/gpu_profile.sh)
    try:
        commit = open(root + "/.git_sha").read().strip()
    except OSError:
        commit = "unknown"
res = {"tag": tag, "commit": commit,
       "bench_args": os.environ.get("BENCH_ARGS", ""),
       "command": "rocprofv3 --pmc <group> --kernel-trace -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras [bench_args], one "
                  "run per counter group (tools/profile_round.sh); durations from the --kernel-trace --stats run of the same command",
       "units": "FETCH_SIZE / WRITE_SIZE in KB as reported (uncorrected: MI355X_MICROARCH.md calibrates the x2 only for 16 B/lane "
                "streaming reads, this path gathers 64-192 B runs); *_frac relative to 8 TB/s HBM, 34.5 TB/s L2, 157.3 TFLOP/s f32 MFMA",
       "kernels": kernels}
res["kernels"]["nmf_vm_query_bwd_segments"] = {
    "note": "the dominant C-ABI call = binning + k_vm_bwd_brick<density> + <appearance>; hbm bytes of the density walk",
    "hbm_bytes_per_launch": walk.get("hbm_bytes_per_launch")}
json.dump(res, open(f"{out}/{tag}_roofline.json", "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters_per_launch"} for k, v in kernels.items()}, indent=1)[:6000])
PY
python "$ROOT/tools/roofline_metrics.py" "$OUT/${TAG}_roofline.json"
