#!/bin/bash
# From the build container: record the commit next to the sources (the GPU box has no .git), then collect the round's rocprofv3
# evidence there:   tools/gpu_profile.sh <tag>     -> gpurun_out/<tag>_*  (copy what is to be judged into profiles/)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
git -C "$ROOT" rev-parse --short HEAD > "$ROOT/.git_sha"
/usr/local/graft/bin/gpurun --timeout 1500 -- "cd /root/repo; bash tools/profile_round.sh $1 > gpurun_out/$1_profile.log 2>&1; tail -5 gpurun_out/$1_profile.log"
