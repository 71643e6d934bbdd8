#!/bin/bash
# usage: tools/prof.sh <tag> <command...>   (run on the GPU box; pass ABSOLUTE script paths)
# One kernel-trace/stats pass, then separate PMC passes (never combined with tracing
# domains), each under its own timeout: a rocprofv3 pass that aborts must not eat the
# GPU budget (a TA_* counter set did exactly that in round 1).
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
PASS_TIMEOUT=${PASS_TIMEOUT:-150}
if [ "${TRACE:-1}" = "1" ]; then
timeout -k 5 $PASS_TIMEOUT rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- "$@" > "$out/trace.log" 2>&1
fi
pi=0
if [ "${PMC:-1}" = "1" ]; then
  for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
              "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
              "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
              "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
              "GRBM_GUI_ACTIVE"; do
    pi=$((pi + 1))
    # PMC_ONLY="3 4": only those passes (1-based)
    if [ -n "${PMC_ONLY:-}" ] && ! echo " $PMC_ONLY " | grep -q " $pi "; then continue; fi
    name=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
    timeout -k 5 $PASS_TIMEOUT rocprofv3 --pmc $pass --kernel-include-regex "msda|bevsca|bevtsa|linear_splitbf16|linear_panel|linear_chain|linear_pipe|wgrad|plan_|layernorm" --output-format csv \
        -d "$out/pmc_$name" -- "$@" > "$out/pmc_$name.log" 2>&1 || echo "pass '$pass' failed/timed out" >> "$out/failed_passes.txt"
  done
fi
cd "$root"
python "$root/tools/prof_summary.py" "$out"
