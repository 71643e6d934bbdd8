# SpatialCrossAttention's chain kernel walking every camera's row (no fold launch) against fold launch + two-row gather, interleaved
run() { BEVMSDA_CHAIN_GATHER_ALL=$1 python bench.py --no-cpu-baseline --no-variants --steps 20 --windows 5 ${@:2} 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('gather_all=$1 [${*:2}] ms_per_step %.4f' % l['ms_per_step'], (l.get('parity') or {}).get('ok'))"; }
for r in 1 2 3; do
  run 1; run 0
done
run 1 --simulate-rank 0,8; run 0 --simulate-rank 0,8; run 1 --simulate-rank 0,8; run 0 --simulate-rank 0,8
run 1 --gemm bf16 --value-storage bf16; run 0 --gemm bf16 --value-storage bf16
run 1 --queue 4; run 0 --queue 4
