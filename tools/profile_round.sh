#!/bin/bash
# Collect the per-round rocprofv3 evidence on an MI355X box:  tools/profile_round.sh <tag>   (e.g. r01_q)
# Writes gpurun_out/<tag>_bench.json, <tag>_bench_kernel_stats.csv, <tag>_pmc_{FETCH,WRITE}_SIZE_per_kernel.csv and
# pmc_vm_bwd.json; copy what is to be judged into profiles/.  PMC passes are separate runs with --kernel-trace only.
set -u
TAG=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
python "$ROOT/bench.py" --steps 30 --warmup 8 > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
rm -rf /tmp/prof_ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_prof_bench.log" 2>&1
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
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/${TAG}_pmc_$C.log" 2>&1
done
python - "$TAG" "$OUT" <<'PY'
import csv, glob, json, sys, collections
tag, out = sys.argv[1], sys.argv[2]
per = {}
walk = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/prof_{c}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    rows = []
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
        if "k_vm_bwd_brick" in k:
            rows.append((k, float(r["Counter_Value"])))
    with open(f"{out}/{tag}_pmc_{c}_per_kernel.csv", "w") as o:
        o.write(f"kernel,launches,{c}_KB_total,{c}_KB_per_launch\n")
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            o.write(f"\"{k}\",{n},{v:.1f},{v / n:.2f}\n")
    per[c] = rows
def avg(rows, pred):
    v = [x for k, x in rows if pred(k)]
    return sum(v) / max(len(v), 1), len(v)
res = {"kernel": "k_vm_bwd_brick<true> (density + normals, all samples) and <false> (appearance, bounce rows)"}
fa, n = avg(per["FETCH_SIZE"], lambda k: True)
wa, _ = avg(per["WRITE_SIZE"], lambda k: True)
res.update(launches_averaged=n, FETCH_SIZE_KB_per_launch=fa, WRITE_SIZE_KB_per_launch=wa,
           hbm_bytes_per_launch=(fa + wa) * 1024, hbm_bytes_per_launch_fetch_x2=(2 * fa + wa) * 1024,
           command="rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline (tools/profile_round.sh)")
kinds = {}
for name, pred in (("density_walk", lambda k: "<true" in k or "true>" in k), ("appearance_walk", lambda k: "<false" in k or "false>" in k)):
    kinds[name] = {"FETCH": avg(per["FETCH_SIZE"], pred)[0], "WRITE": avg(per["WRITE_SIZE"], pred)[0]}
res["per_kind_KB"] = kinds
json.dump(res, open(f"{out}/pmc_vm_bwd.json", "w"), indent=1)
print(json.dumps(res))
PY
tail -1 "$OUT/${TAG}_bench.json"
