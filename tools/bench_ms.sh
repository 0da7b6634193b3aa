#!/bin/bash
# prints ms_per_step of one `python bench.py "$@"` run (A/B helper: `VAR=1 tools/bench_ms.sh --steps 100 --warmup 20 --no-extras --no-cpu-baseline`)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python "$ROOT/bench.py" "$@" 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
