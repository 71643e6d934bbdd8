#!/bin/bash
# A/B of the grad_value sort kernel's walk (round 6): 8-lane groups + LDS stage (default) against the half-wave walk
# (libbevmsda_walk32.so) and the no-atomic diagnostic builds of both.  GPU box:  tools/gv_walk_ab.sh > gpurun_out/r6f_gv_walk_ab.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in ${LIBS:-default walk32 walk8na walk32na}; do
  if [ $lib = default ]; then unset BEVMSDA_LIBRARY; else export BEVMSDA_LIBRARY=$root/bevformer_amd/lib/libbevmsda_$lib.so; fi
  out=$root/gpurun_out/gvwalk_$lib
  rm -rf $out
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python $root/tools/kbench2.py bwd > $out.log 2>&1
  echo "== $lib: operator backward (tools/kbench2.py, padded base operands; rocprofv3 kernel statistics)"
  grep "check" $out.log | cut -c1-200
  python - $out <<'P'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gradvalue_sort" in r["Name"] or "gradloc" in r["Name"]:
            print("   %-95s calls %4s avg %8.1f us min %8.1f max %8.1f" % (r["Name"][:95], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
P
done
cd $root
for lib in ${LIBS2:-default walk32 default walk32}; do
  if [ $lib = default ]; then unset BEVMSDA_LIBRARY; else export BEVMSDA_LIBRARY=$root/bevformer_amd/lib/libbevmsda_$lib.so; fi
  for wl in base small4; do
    extra=""; [ $wl = small4 ] && extra="--gemm bf16 --value-storage bf16"
    python bench.py --workload $wl $extra --no-cpu-baseline --no-variants --backward --steps 10 --warmup 3 --windows 3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$lib $wl $extra  fwd+bwd graph replay ms_per_step %.3f' % d['ms_per_step'])"
  done
done
