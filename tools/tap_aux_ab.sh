#!/bin/bash
# A/B of the cache-policy bits on the sampling kernels' tap loads (library variants built with -DBEVMSDA_TAP_AUX=..): the base
# forward step and its SCA / TSA sampling kernels by HIP events.  tools/tap_aux_ab.sh  (GPU box)
for lib in ${LIBS:-default aux1 aux2 aux16 aux3 default}; do
  if [ $lib = default ]; then unset BEVMSDA_LIBRARY; else export BEVMSDA_LIBRARY=$PWD/bevformer_amd/lib/libbevmsda_$lib.so; fi
  python bench.py --no-cpu-baseline --no-variants --steps 20 --warmup 3 --windows 3 2>/dev/null | python -c "
import json,sys
ls=[l for l in sys.stdin if l.startswith('{')]
d=json.loads(ls[0]); d=d.get('bench_detail', d)
k=d['kernels']
print('$lib', ' ms_per_step %.3f' % d['ms_per_step'], ' sca_fwd %.1f us' % k['sca_fwd']['avg_us'], ' tsa_fwd %.1f us' % k['tsa_fwd']['avg_us'])"
done
