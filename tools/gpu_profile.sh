#!/bin/bash
# From the build container: record the commit next to the sources (the GPU box has no .git), then collect the round's rocprofv3
# evidence there:   tools/gpu_profile.sh <tag>     -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
git -C "$ROOT" rev-parse --short HEAD > "$ROOT/.git_sha"
# tools/gpu_profile.sh <tag> [bench.py args of another regime, e.g. --grid 300]
TAG=$1; shift
/usr/local/graft/bin/gpurun --timeout 1500 -- "cd /root/repo; BENCH_ARGS='$*' bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1; tail -5 gpurun_out/${TAG}_profile.log"
