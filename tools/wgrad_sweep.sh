#!/bin/bash
# Workgroup-target sweep of the multi-problem weight gradient inside the base forward + backward step (eager, HIP-event
# brackets per tag): tools/wgrad_sweep.sh   (GPU box)
for wgs in 0 512 768 1024 1536; do
  echo "== BEVMSDA_WGRAD_WGS=$wgs"
  BEVMSDA_WGRAD_WGS=$wgs python bench.py --no-cpu-baseline --no-variants --backward --graph off --steps 4 --warmup 2 --windows 2 ${WORKLOAD_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ms_per_step %.3f' % d['ms_per_step'])
g=d['gemms']['per_tag']
tot=0
for k,v in sorted(g.items()):
    if k.endswith('_dw') or '_dw' in k:
        print('   %-34s %7.1f us x%d' % (k, v['avg_us'], v['launches'])); tot+=v['avg_us']*v['launches']
print('   weight gradients total %.0f us over the timed launches; all gemm tags %.0f us/step' % (tot, d['gemms']['total_us_per_step']))
"
done
