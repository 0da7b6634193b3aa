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
# (the line is the compact one; the per-kernel table and every leg are in the detail file it names)
cp "$ROOT/bench_detail.json" "$OUT/${TAG}_bench_detail.json" 2>/dev/null
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
python "$ROOT/tools/roofline_metrics.py" --collect "$TAG" "$OUT" "$ROOT" /tmp/prof_pmc_ /tmp/prof_ks || echo "roofline_metrics: some kernels were rejected (see the output above)"
