#!/bin/bash
# A/B on one box, interleaved: the forward step with TSA's [history ; queries] value projected from its two tensors
# (bevmsda_linear_panel_rows2_f32) vs from the stacked copy.   tools/stack_free_ab.sh   (GPU box)
for rep in 1 2 3; do
  for sf in 1 0; do
    BEVMSDA_STACK_FREE=$sf python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 5 --windows 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('stack_free=$sf  ms_per_step %.3f  tsa_value_proj %.1f us' % (d['ms_per_step'], d['gemms']['per_tag']['tsa_value_proj']['avg_us']))"
  done
done
