# A/B of the layer-to-layer seam (BEVMSDA_TSA_SEAM) x the chain kernels' workgroup shape (BEVMSDA_CHAIN_SHAPE) on the default bench
for r in 1 2; do
for cfg in ${CFGS:-"1,0 1,2 0,0 0,2"}; do
  IFS=, read a b <<< "$cfg"
  BEVMSDA_TSA_SEAM=$a BEVMSDA_CHAIN_SHAPE=$b python bench.py --no-cpu-baseline --no-variants --steps 20 --windows 5 --detail-json /tmp/d.json $BENCH_ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read())
d=json.load(open('/tmp/d.json')); d=d.get('bench_detail',d)
pt=d['gemms']['per_tag']
print('seam=$a shape=$b ms_per_step %.4f' % l['ms_per_step'], {k: round(v['avg_us'],1) for k,v in pt.items()}, 'parity', (d.get('parity') or {}).get('ok'))"
done
done
