#!/bin/bash
# usage: tools/prof.sh <tag> <command...>   (run on the GPU box, from the repo root)
# kernel-trace stats + separate PMC passes (never combined with tracing domains).
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- "$@" > "$out/trace.log" 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCC_REQ_sum TCC_ATOMIC_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE"; do
  name=$(echo "$pass" | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d "$out/pmc_$name" -- "$@" > "$out/pmc_$name.log" 2>&1
done
cd "$root"
# compact per-kernel summaries
python - "$out" <<'PY'
import csv, glob, sys, os, collections
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", f)
    for i, row in enumerate(csv.reader(open(f))):
        if i < 12: print(",".join(row))
for d in sorted(glob.glob(out + "/pmc_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
            agg[k][0] += float(row.get("Counter_Value", 0)); agg[k][1] += 1
        print("== pmc", os.path.basename(os.path.dirname(d)))
        for (k, c), (v, n) in sorted(agg.items()):
            if "msda" in k or "bevsca" in k or "bevtsa" in k:
                print(f"{k:60s} {c:40s} total={v:.4g} n={n} per_dispatch={v/max(n,1):.6g}")
PY
