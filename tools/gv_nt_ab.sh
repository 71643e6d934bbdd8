#!/bin/bash
# A/B: non-temporal operand reads in the grad_value sort kernel (libbevmsda_gvnt.so, -DBEVMSDA_GV_STREAM_NT=1): operator time and
# the L2 / fabric counters of the kernel on the image-ordered base SCA operands.   tools/gv_nt_ab.sh   (GPU box)
root=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in ${LIBS:-default gvnt default gvnt}; do
  if [ $lib = default ]; then unset BEVMSDA_LIBRARY; else export BEVMSDA_LIBRARY=$root/bevformer_amd/lib/libbevmsda_$lib.so; fi
  echo "== $lib"; cd $root; python tools/gv_rows_ab.py 2>&1 | grep "default"
  out=$root/gpurun_out/gvnt_$lib; rm -rf $out; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
  i=0
  for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_ATOMIC_sum TCC_REQ_sum"; do
    i=$((i+1))
    timeout -k 5 120 rocprofv3 --pmc $pass --kernel-include-regex "gradvalue_sort" --output-format csv -d $out/p$i -- python $root/tools/gv_one.py 6 sca_image > $out/p$i.log 2>&1 || echo "pass $i failed"
  done
  python - $out <<'P'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]][0] += float(r["Counter_Value"]); agg[r["Counter_Name"]][1] += 1
for k, (v, n) in sorted(agg.items()):
    print("   %-22s %14.0f per launch" % (k, v / n))
if "FETCH_SIZE" in agg and "WRITE_SIZE" in agg:
    f = agg["FETCH_SIZE"][0] / agg["FETCH_SIZE"][1]; w = agg["WRITE_SIZE"][0] / agg["WRITE_SIZE"][1]
    print("   HBM-side bytes: read %.1f MB + written %.1f MB" % (f * 1024 * 2 / 1e6, w * 1024 / 1e6))
P
done
