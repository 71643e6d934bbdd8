#!/bin/bash
# Interleaved on/off A/B of ONE environment switch on the FORWARD step (base, graph replay):  tools/fwd_knob_ab.sh BEVMSDA_CHAIN_NEXT [reps]
knob=$1; reps=${2:-3}
for rep in $(seq $reps); do
  for v in 1 0; do
    env $knob=$v python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 5 --windows 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$knob=$v  forward ms_per_step %.3f  parity %s' % (d['ms_per_step'], (d.get('parity') or {}).get('ok')))"
  done
done
