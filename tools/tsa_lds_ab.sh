# A/B of TemporalSelfAttention's sampling kernel: default (tap lines from the vector L1) vs the LDS-tile forms (BEVMSDA_FUSED_SPEC=5)
python -m pytest tests/test_baseline_configs_gpu.py -q -m gpu -k "staged_in_lds or fused_tsa" -x 2>&1 | tail -5
python tools/tsa_lds_kernel_ab.py 2>&1 | grep -v amdgpu.ids
for r in 1 2; do
for spec in 0 5; do
  BEVMSDA_FUSED_SPEC=$spec python bench.py --no-cpu-baseline --no-variants --steps 20 --windows 5 --detail-json /tmp/d.json 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read())
d=json.load(open('/tmp/d.json')); d=d.get('bench_detail',d)
print('fused_spec=$spec ms_per_step %.4f' % l['ms_per_step'], {k: round(v['avg_us'],1) for k,v in d['kernels'].items()}, 'parity', (d.get('parity') or {}).get('ok'), (d.get('parity') or {}).get('max_abs'))"
done
done
