#!/bin/bash
# A/B of the point-group keying of the grad_value sort kernel (library variants built with -DBEVMSDA_GV_GROUPS_*): the
# operator backward's HIP-event time inside the base forward + backward step.   tools/gv_groups_ab.sh  (GPU box)
for lib in ${LIBS:-default exp_tsa4 exp_sca2 exp_sca1 default}; do
  if [ $lib = default ]; then unset BEVMSDA_LIBRARY; else export BEVMSDA_LIBRARY=$PWD/bevformer_amd/lib/libbevmsda_$lib.so; fi
  python bench.py --no-cpu-baseline --no-variants --backward --graph off --steps 3 --warmup 2 --windows 1 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k=d['kernels']
print('$lib', '  sca_bwd %.1f us' % k['sca_bwd']['avg_us'], ' tsa_bwd %.1f us' % k['tsa_bwd']['avg_us'])"
  python bench.py --no-cpu-baseline --no-variants --backward --steps 5 --warmup 2 --windows 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('      graph replay ms_per_step %.3f' % d['ms_per_step'])"
done
