#!/bin/bash
# launch-by-launch timeline of the LAST training step of a short bench run (rocprofv3 --kernel-trace):
#   tools/step_timeline.sh [bench.py args...]  -> gpurun_out/step_timeline.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/gpurun_out"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ktl
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktl -- python "$ROOT/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --no-extras "$@" > /tmp/ktl.log 2>&1
f=$(find /tmp/ktl -name '*kernel_trace.csv' | head -1)
python - "$f" > "$ROOT/gpurun_out/step_timeline.txt" <<'PY'
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "k_adam" in r["Kernel_Name"]]
lo, hi = adam[-2] + 1, adam[-1] + 1
t0, prev_end, busy = int(rows[lo]["Start_Timestamp"]), None, 0
print(f"{'start_us':>9s} {'gap_us':>7s} {'dur_us':>8s} {'grid':>9s} {'wg':>5s} {'lds':>6s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} scr  kernel")
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"\(.*", "", n)[:70]
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    busy += e - s
    prev_end = e
    g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    w = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)))
    print(f"{(s - t0) / 1e3:9.1f} {gap:7.1f} {(e - s) / 1e3:8.1f} {g:9d} {w:5d} {r.get('LDS_Block_Size', ''):>6s} {r.get('VGPR_Count', ''):>5s} "
          f"{r.get('Accum_VGPR_Count', ''):>5s} {r.get('SGPR_Count', ''):>5s} {r.get('Scratch_Size', ''):>3s} q{r.get('Queue_Id', '')} {n}")
print(f"# span {(prev_end - t0) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us, {hi - lo} launches")
PY
tail -n 3 "$ROOT/gpurun_out/step_timeline.txt"
