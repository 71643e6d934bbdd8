#!/bin/bash
# Round-4 GPU-box visit: tools/gpu_r4.sh <tag> [steps...]   (run from the repo root on the GPU box)
set -u
tag=${1:-r4a}; shift || true
steps=${*:-"pytest smoke bench gloo2"}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd "$root"
nproc > "$out/nproc.txt"
for s in $steps; do
  t0=$(date +%s)
  case $s in
    pytest) timeout 900 python -m pytest tests -m gpu -q -x > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"; tail -6 "$out/pytest_gpu.log";;
    pytest_all) timeout 900 python -m pytest tests -m gpu -q > "$out/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_gpu.log"; tail -30 "$out/pytest_gpu.log";;
    pytest_new) timeout 600 python -m pytest ${PYTEST_FILES:-tests/test_train_path_gpu.py} -q > "$out/pytest_new.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_new.log"; tail -40 "$out/pytest_new.log";;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; tail -3 "$out/smoke.log";;
    bench) timeout 900 python bench.py > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?" >> "$out/bench.err"; python tools/bench_digest.py "$out/bench.json" 2>/dev/null | head -60; tail -3 "$out/bench.err";;
    bench_nov) timeout 600 python bench.py --no-variants > "$out/bench_nov.json" 2> "$out/bench_nov.err"; echo "bench rc=$?" >> "$out/bench_nov.err"; cut -c1-600 "$out/bench_nov.json"; tail -3 "$out/bench_nov.err";;
    gloo2) timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --dist-backend gloo --steps 5 --warmup 2 --windows 2 > "$out/bench_gloo2.json" 2> "$out/bench_gloo2.err"; echo "rc=$?" >> "$out/bench_gloo2.err"; tail -c 3000 "$out/bench_gloo2.json"; tail -5 "$out/bench_gloo2.err";;
    tile1) timeout 300 python bench.py --no-cpu-baseline --graph on --force-tiling --no-variants > "$out/bench_tile1.json" 2> "$out/bench_tile1.err"; cut -c1-400 "$out/bench_tile1.json"; tail -3 "$out/bench_tile1.err";;
    bwd_base) timeout 400 python bench.py --no-cpu-baseline --no-variants --backward --steps 5 --warmup 2 --windows 3 ${BWD_ARGS:-} > "$out/bench_bwd_base.json" 2> "$out/bench_bwd_base.err"; echo "rc=$?" >> "$out/bench_bwd_base.err"; cut -c1-500 "$out/bench_bwd_base.json"; tail -3 "$out/bench_bwd_base.err";;
    bwd_small4) timeout 400 python bench.py --no-cpu-baseline --no-variants --backward --workload small4 --steps 5 --warmup 2 --windows 3 ${BWD_ARGS:-} > "$out/bench_bwd_small4.json" 2> "$out/bench_bwd_small4.err"; echo "rc=$?" >> "$out/bench_bwd_small4.err"; cut -c1-500 "$out/bench_bwd_small4.json"; tail -3 "$out/bench_bwd_small4.err";;
    bwd_small4_bf16) timeout 400 python bench.py --no-cpu-baseline --no-variants --backward --workload small4 --gemm bf16 --value-storage bf16 --steps 5 --warmup 2 --windows 3 ${BWD_ARGS:-} > "$out/bench_bwd_small4_bf16.json" 2> "$out/bench_bwd_small4_bf16.err"; echo "rc=$?" >> "$out/bench_bwd_small4_bf16.err"; cut -c1-500 "$out/bench_bwd_small4_bf16.json"; tail -3 "$out/bench_bwd_small4_bf16.err";;
    trace_bwd) PMC=0 PASS_TIMEOUT=240 timeout 300 tools/prof.sh "${tag}_bwd${TRACE_TAG:-}" python "$root/bench.py" --no-cpu-baseline --no-variants --backward --graph off --steps 3 --warmup 1 --windows 1 ${BWD_ARGS:-} > "$out/prof_bwd_summary.txt" 2>&1; head -45 "$out/prof_bwd_summary.txt" | cut -c1-170;;
    trace_fwd) PMC=0 PASS_TIMEOUT=240 timeout 300 tools/prof.sh "${tag}_fwd" python "$root/bench.py" --no-cpu-baseline --no-variants --graph off --steps 3 --warmup 1 --windows 1 > "$out/prof_fwd_summary.txt" 2>&1; head -30 "$out/prof_fwd_summary.txt" | cut -c1-170;;
    trace_sim) PMC=0 PASS_TIMEOUT=240 timeout 300 tools/prof.sh "${tag}_sim" python "$root/bench.py" --no-cpu-baseline --no-variants --simulate-rank ${SIM_RANK:-3,8} --tile-layout ${SIM_LAYOUT:-sectors} --graph off --steps 3 --warmup 1 --windows 1 > "$out/prof_sim_summary.txt" 2>&1; head -45 "$out/prof_sim_summary.txt" | cut -c1-170;;
    bench_sim) timeout 300 python bench.py --no-cpu-baseline --no-variants --simulate-rank ${SIM_RANK:-3,8} --tile-layout ${SIM_LAYOUT:-sectors} --steps 10 --warmup 3 --windows 3 > "$out/bench_sim.json" 2> "$out/bench_sim.err"; echo "rc=$?" >> "$out/bench_sim.err"; cut -c1-400 "$out/bench_sim.json"; tail -3 "$out/bench_sim.err";;
    traffic_fwd) timeout 900 python tools/profile_traffic.py --config base_fwd --tag "$tag" > "$out/traffic_fwd.log" 2>&1; tail -15 "$out/traffic_fwd.log";;
    traffic_bwd) timeout 900 python tools/profile_traffic.py --config base_bwd --tag "$tag" > "$out/traffic_bwd.log" 2>&1; tail -15 "$out/traffic_bwd.log";;
    custom) timeout ${CUSTOM_TIMEOUT:-600} bash -c "${CUSTOM_CMD}" > "$out/custom.log" 2>&1; echo "rc=$?" >> "$out/custom.log"; tail -${CUSTOM_TAIL:-60} "$out/custom.log";;
  esac
  echo "== step $s took $(( $(date +%s) - t0 )) s"
done
