#!/bin/bash
# per-kernel time of a short bench run:  tools/kstats.sh [bench.py args...]   (rocprofv3 --kernel-trace --stats)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/kstats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-extras "$@" > /tmp/kstats.log 2>&1
f=$(find /tmp/kstats -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    n = re.sub(r"\(.*", "", n)[:52]
    print(f"{n:54s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:9.1f} us  {r['Percentage']:>6s} %")
PY
