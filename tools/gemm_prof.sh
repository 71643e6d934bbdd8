#!/bin/bash
# usage: tools/gemm_prof.sh <tag>   (on the GPU box) — SQ / LDS / cache counters of the projection
# GEMM kernel at two layer shapes, one rocprofv3 --pmc pass per counter group (never combined
# with tracing), each under its own timeout; unknown counter names only lose their own pass.
set -u
tag=${1:-gemm}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
CMD="python $root/tools/gbench.py --only output_proj,sca_value_proj --modes ${MODES:-split} --iters 4"
i=0
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" \
            "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
            "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
            "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum" \
            "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 ${PASS_TIMEOUT:-90} rocprofv3 --pmc $pass --kernel-include-regex "linear_splitbf16" --output-format csv \
      -d "$out/pmc_$i" -- $CMD > "$out/pmc_$i.log" 2>&1 || echo "pass $i '$pass' failed/timed out" >> "$out/failed_passes.txt"
done
cd "$root"
PROF_BY_GRID=1 python "$root/tools/prof_summary.py" "$out" > "$out/summary.txt" 2>&1
tail -c 6000 "$out/pmc.json"
cat "$out/failed_passes.txt" 2>/dev/null
