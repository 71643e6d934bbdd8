#!/bin/bash
# Interleaved on/off A/B of ONE environment switch on one box, graph-replayed fwd + bwd step at base and at configs[2]:
#   tools/knob_ab.sh BEVMSDA_FUSED_SAVE [reps]      (GPU box)
knob=$1; reps=${2:-3}
for rep in $(seq $reps); do
  for v in 1 0; do
    for wl in "" "--workload small4 --gemm bf16 --value-storage bf16"; do
      env $knob=$v python bench.py --no-cpu-baseline --no-variants --backward --steps 5 --warmup 2 --windows 3 $wl 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$knob=$v %-12s ms_per_step %.3f' % ('small4_bf16' if '$wl' else 'base', d['ms_per_step']))"
    done
  done
done
