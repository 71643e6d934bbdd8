#!/bin/bash
# usage: tools/trace_cmd.sh <tag> <command...> : rocprofv3 kernel trace + stats of a command, top kernels printed
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- "$@" > "$out/trace.log" 2>&1
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
cp "$f" "$out/kernel_stats.csv" 2>/dev/null
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:110]:110s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']}%")
PY
