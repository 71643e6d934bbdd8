# per-tag time of the hoisted value projections by row-panel shape (BEVMSDA_GEMM_KERNEL: default = the library's rule)
for r in 1 2; do
for k in "" panel64 panel128 panel64w2 panel64w6; do
  BEVMSDA_GEMM_KERNEL=$k python bench.py --no-cpu-baseline --no-variants --steps 10 --windows 3 --detail-json /tmp/d.json 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read())
d=json.load(open('/tmp/d.json')); d=d.get('bench_detail',d)
pt=d['gemms']['per_tag']
print('kernel=${k:-default} ms_per_step %.4f' % l['ms_per_step'], {k: round(v['avg_us'],1) for k,v in pt.items()})"
done
done
