#!/bin/bash
# Interleaved A/B of two training-path switches on one box (graph-replayed fwd + bwd step):  tools/train_ab.sh  (GPU box)
#   BEVMSDA_WEIGHT_VIEWS   W^T weight images packed from W  vs  a contiguous transpose per weight and step
#   BEVMSDA_FLATTEN_PARAMS merged projections' parameters back to back (views)  vs  torch.cat per group and step
for rep in 1 2; do
  for cfg in "1 1" "0 1" "1 0" "0 0"; do
    set -- $cfg
    for wl in "" "--workload small4 --gemm bf16 --value-storage bf16"; do
      BEVMSDA_WEIGHT_VIEWS=$1 BEVMSDA_FLATTEN_PARAMS=$2 python bench.py --no-cpu-baseline --no-variants --backward --steps 5 --warmup 2 --windows 3 $wl 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('views=$1 flatten=$2 %-12s ms_per_step %.3f' % ('small4_bf16' if '$wl' else 'base', d['ms_per_step']))"
    done
  done
done
