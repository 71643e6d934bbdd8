#!/bin/bash
# What the training forward keeps for its backward (BEVMSDA_FUSED_SAVE): 2 = attention weights only, locations recomputed by the
# backward kernels (round 6); 1 = locations + weights; 0 = nothing (expand pass).  Interleaved on one box, base step + configs[2].
for v in ${MODES:-2 1 0 2 1 0}; do
  export BEVMSDA_FUSED_SAVE=$v
  python bench.py --no-cpu-baseline --no-variants --backward --steps 10 --warmup 3 --windows 3 2>/dev/null | python -c "
import json,sys
ls=[l for l in sys.stdin if l.startswith('{')]
d=json.loads(ls[0]); d=d.get('bench_detail', d)
k=d['kernels']
print('fused_save=$v base fwd+bwd ms_per_step %.3f' % d['ms_per_step'], {t: round(k[t]['avg_us'],1) for t in k})"
  python bench.py --workload small4 --gemm bf16 --value-storage bf16 --no-cpu-baseline --no-variants --backward --steps 10 --warmup 3 --windows 3 2>/dev/null | python -c "
import json,sys
ls=[l for l in sys.stdin if l.startswith('{')]
d=json.loads(ls[0]); d=d.get('bench_detail', d)
print('fused_save=$v small4 bf16 fwd+bwd ms_per_step %.3f' % d['ms_per_step'])"
done
