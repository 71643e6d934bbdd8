#!/bin/bash
# SQ counters of the grad_value sort kernel on the image-ordered base SCA operator: tools/gv_pmc.sh <tag> [lib]
tag=$1; lib=${2:-default}
root=${GRAFT_REPO_ROOT:-$(pwd)}
if [ $lib != default ]; then export BEVMSDA_LIBRARY=$root/bevformer_amd/lib/libbevmsda_$lib.so; fi
out=$root/gpurun_out/gvpmc_$tag
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_IFETCH" \
            "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $pass --kernel-include-regex "gradvalue_sort" --output-format csv -d $out/p$i -- python $root/tools/gv_one.py 6 ${CASE:-sca_image} > $out/p$i.log 2>&1 || echo "pass $i failed" >> $out/failed.txt
done
python - $out <<'P'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Counter_Name"]
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for k, (v, n) in sorted(agg.items()):
    print("%-28s %16.0f per dispatch (%d dispatches)" % (k, v / n, n))
P
