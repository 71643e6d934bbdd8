# A/B of the chain kernels' workgroup shape (BEVMSDA_CHAIN_SHAPE: 0 = the library's rule, 2 = 32-row workgroups everywhere) on the
# training step (the saving forms of the chain kernels write 5 KB per row more than the inference forms)
for r in 1 2; do
for sh in 0 2; do
  BEVMSDA_CHAIN_SHAPE=$sh python bench.py --backward --no-cpu-baseline --no-variants --steps 10 --windows 5 --detail-json /tmp/d.json 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('base fwd+bwd shape=$sh ms_per_step %.3f' % l['ms_per_step'])"
  BEVMSDA_CHAIN_SHAPE=$sh python bench.py --backward --workload small4 --gemm bf16 --value-storage bf16 --no-cpu-baseline --no-variants --steps 10 --windows 5 --detail-json /tmp/d.json 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('small4 bf16 fwd+bwd shape=$sh ms_per_step %.3f' % l['ms_per_step'])"
done
done
python bench.py --force-tiling --no-cpu-baseline --no-variants --steps 10 --windows 3 2>&1 | tail -1 | cut -c1-600
