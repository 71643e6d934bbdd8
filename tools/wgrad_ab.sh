#!/bin/bash
# A/B of the two weight-gradient kernels inside the forward + backward step (GPU box): tools/wgrad_ab.sh
for round in 1 2; do
for v in 1 0; do
  echo "== BEVMSDA_WGRAD_VARIANT=$v (0 = bf16 planes + transposing LDS reads, 1 = gathered fragments) ${WORKLOAD_ARGS:-}"
  BEVMSDA_WGRAD_VARIANT=$v python bench.py --no-cpu-baseline --no-variants --backward --steps 5 --warmup 2 --windows 3 ${WORKLOAD_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('   graph replay: ms_per_step %.3f' % d['ms_per_step'])"
  BEVMSDA_WGRAD_VARIANT=$v python bench.py --no-cpu-baseline --no-variants --backward --graph off --steps 3 --warmup 2 --windows 1 ${WORKLOAD_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
g=d['gemms']['per_tag']
tot=0
for k,v in sorted(g.items()):
    if '_dw' in k:
        print('   %-34s %7.1f us x%d' % (k, v['avg_us'], v['launches'])); tot+=v['avg_us']*v['launches']/3
print('   weight gradients %.0f us per step' % tot)"
done; done
