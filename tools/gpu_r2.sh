#!/bin/bash
# One GPU-box visit of round 2.  usage: tools/gpu_r2.sh <tag> [steps...]
set -u
tag=${1:-r2}; shift || true
steps=${*:-"pytest bench"}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
for s in $steps; do
  case $s in
    pytest) timeout 1500 python -m pytest tests -m gpu -q --durations=12 > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"; grep -E "passed|failed|FAILED|ERROR|rc=" "$out/pytest_gpu.log" | tail -40;;
    pytest_new) timeout 1200 python -m pytest tests/test_frame_plan_gpu.py tests/test_baseline_configs_gpu.py tests/test_prologue_gpu.py -m gpu -q --durations=8 > "$out/pytest_new.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_new.log"; grep -E "passed|failed|FAILED|ERROR|rc=|Error|assert" "$out/pytest_new.log" | tail -60;;
    bench) timeout 900 python bench.py > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/bench.err"; tail -3 "$out/bench.err"; python tools/bench_digest.py "$out/bench.json";;
    bench_quick) timeout 600 python bench.py --no-cpu-baseline > "$out/bench_quick.json" 2> "$out/bench_quick.err"; echo "rc=$?" >> "$out/bench_quick.err"; tail -3 "$out/bench_quick.err"; python tools/bench_digest.py "$out/bench_quick.json";;
    bench_polar) timeout 600 python bench.py --no-cpu-baseline --no-variants --row-order polar > "$out/bench_polar.json" 2> "$out/bench_polar.err"; tail -2 "$out/bench_polar.err"; python tools/bench_digest.py "$out/bench_polar.json";;
    bench_main) timeout 600 python bench.py --no-cpu-baseline --no-variants > "$out/bench_main.json" 2> "$out/bench_main.err"; tail -2 "$out/bench_main.err"; python tools/bench_digest.py "$out/bench_main.json";;
    bench_static) timeout 600 python bench.py --no-cpu-baseline --no-variants --static-rig > "$out/bench_static.json" 2> "$out/bench_static.err"; tail -2 "$out/bench_static.err"; python tools/bench_digest.py "$out/bench_static.json";;
    bench_hostplans) timeout 600 python bench.py --no-cpu-baseline --no-variants --host-plans > "$out/bench_hostplans.json" 2> "$out/bench_hostplans.err"; tail -2 "$out/bench_hostplans.err"; python tools/bench_digest.py "$out/bench_hostplans.json";;
    bench_eager) timeout 600 python bench.py --no-cpu-baseline --graph off > "$out/bench_eager.json" 2> "$out/bench_eager.err"; tail -2 "$out/bench_eager.err"; python tools/bench_digest.py "$out/bench_eager.json";;
    bench_bwd) timeout 600 python bench.py --no-cpu-baseline --backward --steps 5 --warmup 2 --windows 3 > "$out/bench_bwd.json" 2> "$out/bench_bwd.err"; tail -2 "$out/bench_bwd.err"; python tools/bench_digest.py "$out/bench_bwd.json";;
    bench_tile1) timeout 600 python bench.py --no-cpu-baseline --force-tiling > "$out/bench_tile1.json" 2> "$out/bench_tile1.err"; tail -2 "$out/bench_tile1.err"; python tools/bench_digest.py "$out/bench_tile1.json";;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; tail -3 "$out/smoke.log";;
    trace) PMC=0 timeout 400 tools/prof.sh "$tag" python "$root/bench.py" --no-cpu-baseline --graph off --steps 3 --warmup 1 --windows 1 > "$out/prof_summary.txt" 2>&1; head -40 "$out/prof_summary.txt" | cut -c1-170;;
    prof) PMC=1 timeout 1200 tools/prof.sh "$tag" python "$root/bench.py" --no-cpu-baseline --graph off --steps 3 --warmup 1 --windows 1 > "$out/prof_summary.txt" 2>&1; tail -30 "$out/prof_summary.txt" | cut -c1-170;;
    prof_bwd) PMC=1 timeout 1200 tools/prof.sh "${tag}_bwd" python "$root/bench.py" --no-cpu-baseline --backward --steps 2 --warmup 1 --windows 1 > "$out/prof_bwd_summary.txt" 2>&1; tail -30 "$out/prof_bwd_summary.txt" | cut -c1-170;;
    trace_sim) PMC=0 timeout 400 tools/prof.sh "${tag}_sim" python "$root/bench.py" --no-cpu-baseline --no-variants --simulate-rank 3,8 --graph off --steps 3 --warmup 1 --windows 1 > "$out/prof_sim_summary.txt" 2>&1; head -40 "$out/prof_sim_summary.txt" | cut -c1-170;;
    trace_bwd) PMC=0 timeout 400 tools/prof.sh "${tag}_bwd" python "$root/bench.py" --no-cpu-baseline --backward --steps 2 --warmup 1 --windows 1 > "$out/prof_bwd_summary.txt" 2>&1; head -40 "$out/prof_bwd_summary.txt" | cut -c1-170;;
    kb*) timeout 600 python tools/kbench2.py > "$out/$s.log" 2>&1; tail -30 "$out/$s.log";;
  esac
done
