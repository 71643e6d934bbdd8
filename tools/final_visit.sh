# The round-end visit: full GPU suite, smoke, default bench (tools/final_visit.sh [traffic] adds the PMC traffic profiles).
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/final/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1
python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
python tools/bench_digest.py gpurun_out/final/bench_default.json > gpurun_out/final/bench_digest.txt 2>&1
if [ "$1" = "traffic" ]; then
  python tools/profile_traffic.py --config base_fwd > gpurun_out/final/traffic_fwd.log 2>&1
  python tools/profile_traffic.py --config base_bwd > gpurun_out/final/traffic_bwd.log 2>&1
  python tools/profile_traffic.py --config small4_bwd_bf16 > gpurun_out/final/traffic_small4.log 2>&1
fi
tail -3 gpurun_out/final/pytest_gpu.log; tail -2 gpurun_out/final/smoke.log; cut -c1-160 gpurun_out/final/bench_digest.txt | head -40
