#!/bin/bash
# Per-kernel counters of any command on the GPU box:   tools/kprof.sh <tag> <kernel-name-substring> <command ...>
# One rocprofv3 --pmc run per counter group (with --kernel-trace only), then a table of per-launch averages for the kernels
# whose name contains the substring -> gpurun_out/<tag>_kprof.txt
set -u
TAG=$1; PAT=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
GROUPS_="SQ_WAVES,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,SQ_INSTS_VALU,SQ_INSTS_MFMA,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_ANY,SQ_WAIT_INST_ANY,SQ_WAIT_ANY,SQ_INST_CYCLES_VMEM,SQ_ACTIVE_INST_LDS,SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_MISC,SQ_ACTIVE_INST_SCA,SQ_INSTS_FLAT,SQ_ACTIVE_INST_FLAT,SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE,GRBM_COUNT"
i=0
for G in $GROUPS_; do
  i=$((i+1))
  rm -rf /tmp/kprof_$i
  (cd "$ROOT" && timeout 300 rocprofv3 --pmc $(echo $G | tr ',' ' ') --kernel-trace --output-format csv -d /tmp/kprof_$i -- "$@") > "$OUT/${TAG}_kprof_$i.log" 2>&1 || echo "counter group $G failed"
done
python - "$PAT" > "$OUT/${TAG}_kprof.txt" <<'PY'
import csv, glob, sys, collections
pat = sys.argv[1]
sums = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(set))
for d in sorted(glob.glob("/tmp/kprof_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if pat not in k:
                continue
            k = k.replace("(anonymous namespace)::", "").split("(")[0][-60:]
            sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]].add(r.get("Dispatch_Id", len(n[k][r["Counter_Name"]])))
for k in sums:
    print(k)
    for c in sorted(sums[k]):
        print(f"   {c:28s} {sums[k][c] / len(n[k][c]):16.1f}   ({len(n[k][c])} launches)")
PY
cat "$OUT/${TAG}_kprof.txt"
