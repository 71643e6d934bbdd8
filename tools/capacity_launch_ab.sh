# SCA sampling launch sized by the row CAPACITY (no tail launch; surplus workgroups return at once) against hint + tail, interleaved
run() { BEVMSDA_FUSED_CAPACITY=$1 python bench.py --no-cpu-baseline --no-variants --steps 20 --windows 5 ${@:2} 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); r=l.get('roofline') or {}; print('capacity=$1 [${*:2}] ms_per_step %.4f  sca avg_us %s' % (l['ms_per_step'], r.get('avg_us')))"; }
for r in 1 2; do
  run 1 --simulate-rank 0,8; run 0 --simulate-rank 0,8
  run 1 --simulate-rank 0,2; run 0 --simulate-rank 0,2
  run 1; run 0
  run 1 --workload tiny; run 0 --workload tiny
done
