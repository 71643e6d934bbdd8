# a simulated rank's step (BEV tiling): plan kernels on the side / main stream, interleaved
run() { BEVMSDA_PLAN_SIDE=$1 python bench.py --no-cpu-baseline --no-variants --steps 20 --windows 5 ${@:2} 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('plan_side=$1 [${*:2}] ms_per_step %.4f' % l['ms_per_step'])"; }
for r in 1 2; do
  run 1 --simulate-rank 0,8; run 0 --simulate-rank 0,8
  run 1 --simulate-rank 3,8; run 0 --simulate-rank 3,8
  run 1 --simulate-rank 0,2; run 0 --simulate-rank 0,2
done
