#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprof summaries, operator sweep.
# usage (from the repo root on the GPU box): tools/gpu_round.sh <tag>
set -u
tag=${1:-r1}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock|gfx" | head -12 > "$out/rocminfo.txt"
nproc > "$out/nproc.txt"
timeout 900 python -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"
timeout 600 python bench.py > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/bench.err"
timeout 300 python tools/kbench.py --quick > "$out/kbench.log" 2>&1
timeout 900 tools/prof.sh "$tag" python "$root/bench.py" --no-cpu-baseline --steps 5 --warmup 2 > "$out/prof_summary.txt" 2>&1
tail -3 "$out/pytest_gpu.log"; cat "$out/bench.json"; tail -5 "$out/kbench.log"
