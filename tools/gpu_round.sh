#!/bin/bash
# One GPU-box visit: parity tests, operator sweep, probes, bench line, rocprof summaries.
# usage (from the repo root on the GPU box): tools/gpu_round.sh <tag> [steps...]
set -u
tag=${1:-r1}; shift || true
steps=${*:-"pytest kbench probe bench prof"}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
nproc > "$out/nproc.txt"
for s in $steps; do
  case $s in
    pytest) timeout 600 python -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"; tail -4 "$out/pytest_gpu.log";;
    kbench) timeout 400 python tools/kbench.py --sweep --sort --iters 10 > "$out/kbench.log" 2>&1; cp gpurun_out/kbench.json "$out/kbench.json" 2>/dev/null; tail -3 "$out/kbench.log";;
    kquick) timeout 300 python tools/kbench.py --iters 10 > "$out/kbench.log" 2>&1; tail -12 "$out/kbench.log";;
    probe) (cd tools/probes && hipcc --offload-arch=gfx950 -O3 -o atomic_probe atomic_probe.hip 2>/dev/null && timeout 120 ./atomic_probe) > "$out/atomic_probe.log" 2>&1; tail -16 "$out/atomic_probe.log";;
    bench) timeout 600 python bench.py > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/bench.err"; cat "$out/bench.json";;
    bench_vbf16) timeout 300 python bench.py --no-cpu-baseline --value-storage bf16 > "$out/bench_vbf16.json" 2> "$out/bench_vbf16.err"; cut -c1-300 "$out/bench_vbf16.json"; tail -2 "$out/bench_vbf16.err";;
    bench_allbf16) timeout 300 python bench.py --no-cpu-baseline --value-storage bf16 --gemm bf16 > "$out/bench_allbf16.json" 2> "$out/bench_allbf16.err"; cut -c1-300 "$out/bench_allbf16.json"; tail -2 "$out/bench_allbf16.err";;
    bench_eager) timeout 300 python bench.py --no-cpu-baseline --graph off > "$out/bench_eager.json" 2> "$out/bench_eager.err"; cut -c1-300 "$out/bench_eager.json"; tail -2 "$out/bench_eager.err";;
    bench_image) timeout 300 python bench.py --no-cpu-baseline --row-order image > "$out/bench_image.json" 2> "$out/bench_image.err"; cat "$out/bench_image.json";;
    bench_raster) timeout 300 python bench.py --no-cpu-baseline --row-order raster > "$out/bench_raster.json" 2> "$out/bench_raster.err"; cat "$out/bench_raster.json";;
    bench_graph) timeout 300 python bench.py --no-cpu-baseline --graph on > "$out/bench_graph.json" 2> "$out/bench_graph.err"; cat "$out/bench_graph.json"; tail -2 "$out/bench_graph.err";;
    bench_tile1) timeout 300 python bench.py --no-cpu-baseline --graph on --force-tiling > "$out/bench_tile1.json" 2> "$out/bench_tile1.err"; cat "$out/bench_tile1.json"; tail -3 "$out/bench_tile1.err";;
    bench_tile1_ff) timeout 300 python bench.py --no-cpu-baseline --graph on --force-tiling --first-frame > "$out/bench_tile1_ff.json" 2> "$out/bench_tile1_ff.err"; cat "$out/bench_tile1_ff.json"; tail -3 "$out/bench_tile1_ff.err";;
    gbench) timeout 300 python tools/gbench.py --iters 20 > "$out/gbench.log" 2>&1; cp gpurun_out/gbench.json "$out/gbench.json" 2>/dev/null; cat "$out/gbench.log" | cut -c1-200;;
    bench_native) timeout 300 python bench.py --no-cpu-baseline --gemm native > "$out/bench_native.json" 2> "$out/bench_native.err"; cut -c1-400 "$out/bench_native.json"; tail -2 "$out/bench_native.err";;
    bench_split) timeout 300 python bench.py --no-cpu-baseline --gemm split > "$out/bench_split.json" 2> "$out/bench_split.err"; cut -c1-400 "$out/bench_split.json"; tail -2 "$out/bench_split.err";;
    bench_split_graph) timeout 300 python bench.py --no-cpu-baseline --gemm split --graph on > "$out/bench_split_graph.json" 2> "$out/bench_split_graph.err"; cut -c1-400 "$out/bench_split_graph.json"; tail -2 "$out/bench_split_graph.err";;
    bench_bf16) timeout 300 python bench.py --no-cpu-baseline --gemm bf16 > "$out/bench_bf16.json" 2> "$out/bench_bf16.err"; cut -c1-400 "$out/bench_bf16.json"; tail -2 "$out/bench_bf16.err";;
    trace_split) PMC=0 timeout 300 tools/prof.sh "$tag" python "$root/bench.py" --no-cpu-baseline --gemm split --steps 3 --warmup 1 > "$out/prof_summary.txt" 2>&1; head -24 "$out/prof_summary.txt" | cut -c1-160;;
    prof_split) PMC=1 timeout 900 tools/prof.sh "$tag" python "$root/bench.py" --no-cpu-baseline --gemm split --steps 3 --warmup 1 > "$out/prof_summary.txt" 2>&1; head -24 "$out/prof_summary.txt" | cut -c1-160;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; tail -3 "$out/smoke.log";;
    prof) PMC=1 timeout 900 tools/prof.sh "$tag" python "$root/bench.py" --no-cpu-baseline --graph off --steps 3 --warmup 1 > "$out/prof_summary.txt" 2>&1; tail -5 "$out/prof_summary.txt";;
    trace) PMC=0 timeout 300 tools/prof.sh "$tag" python "$root/bench.py" --no-cpu-baseline --graph off --steps 3 --warmup 1 > "$out/prof_summary.txt" 2>&1; head -30 "$out/prof_summary.txt";;
  esac
done
