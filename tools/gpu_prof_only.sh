#!/bin/bash
# rocprofv3 kernel-trace + PMC passes of the default bench workload (GPU box).
set -u
tag=${1:-prof}
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out/$tag"
cd "$root"
timeout 1200 tools/prof.sh "$tag" python "$root/bench.py" --no-cpu-baseline --steps 5 --warmup 2 > "$root/gpurun_out/$tag/prof_summary.txt" 2>&1
tail -60 "$root/gpurun_out/$tag/prof_summary.txt"
