# SCA sampling by query-tile size (rows of one head in adjacent lane groups): library variants built with -DBEVMSDA_QTILE_FWD=N
for v in default 4 16 32 128; do
  lib=$PWD/bevformer_amd/lib/libbevmsda_qt$v.so; [ $v = default ] && lib=$PWD/bevformer_amd/lib/libbevmsda.so
  BEVMSDA_LIBRARY=$lib python bench.py --no-cpu-baseline --no-variants --steps 10 --windows 3 --detail-json /tmp/d.json 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read())
d=json.load(open('/tmp/d.json')); d=d.get('bench_detail',d)
print('qtile=$v ms_per_step %.4f' % l['ms_per_step'], {k: round(v['avg_us'],1) for k,v in d['kernels'].items()}, 'parity', (d.get('parity') or {}).get('ok'))"
done
