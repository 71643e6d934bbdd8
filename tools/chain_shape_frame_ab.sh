# seam kernels' workgroup shape at FRAME level (replayed graph): default (mixed: whole rounds of 64-row panels + a 32-row tail launch)
# against 32-row panels in one launch (BEVMSDA_CHAIN_SHAPE=2), interleaved
run() { BEVMSDA_CHAIN_SHAPE=$1 python bench.py --no-cpu-baseline --no-variants --steps 20 --windows 5 ${@:2} 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('chain_shape=$1 [${*:2}] ms_per_step %.4f' % l['ms_per_step'])"; }
for r in 1 2 3; do
  run 2; run 0
done
run 2 --gemm bf16 --value-storage bf16; run 0 --gemm bf16 --value-storage bf16
run 2 --gemm bf16 --value-storage bf16; run 0 --gemm bf16 --value-storage bf16
run 2 --queue 4; run 0 --queue 4
run 2 --first-frame; run 0 --first-frame
run 2 --workload small4; run 0 --workload small4
run 2 --backward; run 0 --backward
run 2 --simulate-rank 0,2; run 0 --simulate-rank 0,2
