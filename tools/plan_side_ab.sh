# frame-plan kernels on the side stream (ahead of the hoisted camera-value projection) against the main stream, interleaved
run() { BEVMSDA_PLAN_SIDE=$1 python bench.py --no-cpu-baseline --no-variants --steps 20 --windows 5 ${@:2} 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('plan_side=$1 [${*:2}] ms_per_step %.4f' % l['ms_per_step'], (l.get('parity') or {}))"; }
for r in 1 2 3; do
  run 1; run 0
done
run 1 --gemm bf16 --value-storage bf16; run 0 --gemm bf16 --value-storage bf16
run 1 --first-frame; run 0 --first-frame
run 1 --workload tiny; run 0 --workload tiny
