#!/bin/bash
# train() mode (dropout active) A/B of the backward chain kernels on one box:  tools/train_mode_ab.sh   (GPU box)
for rep in 1 2; do
  for v in 1 0; do
    BEVMSDA_CHAIN_BWD=$v python bench.py --no-cpu-baseline --no-variants --backward --train-mode --steps 5 --warmup 2 --windows 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('train() mode  BEVMSDA_CHAIN_BWD=$v  base ms_per_step %.3f' % d['ms_per_step'])"
  done
done
