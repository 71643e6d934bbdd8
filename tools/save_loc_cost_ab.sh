# What do the location stores cost the training forward?  default vs the diagnostic build without them (tools/build_variant.sh nosaveloc bevmsda_capi.hip -DBEVMSDA_DIAG_NO_SAVE_LOC=1; its gradients are wrong by construction).  GPU box.
for lib in default nosaveloc default nosaveloc; do
  if [ $lib = default ]; then unset BEVMSDA_LIBRARY; else export BEVMSDA_LIBRARY=$PWD/bevformer_amd/lib/libbevmsda_$lib.so; fi
  python bench.py --no-cpu-baseline --no-variants --backward --steps 10 --warmup 3 --windows 3 2>/dev/null | python -c "
import json,sys
ls=[l for l in sys.stdin if l.startswith('{')]
d=json.loads(ls[0]); d=d.get('bench_detail', d)
k=d['kernels']
print('$lib fwd+bwd base ms_per_step %.3f' % d['ms_per_step'], {t: round(k[t]['avg_us'],1) for t in k})"
done
