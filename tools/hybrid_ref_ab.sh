# TemporalSelfAttention's two-entry anchors from one elementwise launch (default) against add + stack + row copy, interleaved
run() { BEVMSDA_HYBRID_REF=$1 python bench.py --no-cpu-baseline --no-variants --steps 20 --windows 5 ${@:2} 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('hybrid_ref=$1 [${*:2}] ms_per_step %.4f' % l['ms_per_step'], (l.get('parity') or {}).get('ok'))"; }
for r in 1 2 3; do
  run 1; run 0
done
run 1 --queue 4; run 0 --queue 4
