# kernel statistics of the replayed base training step with and without the one-launch image rebuild
for v in 1 0; do
  BEVMSDA_IMAGE_BATCH=$v bash tools/trace_cmd.sh ib_trace_$v python $GRAFT_REPO_ROOT/bench.py --no-variants --no-cpu-baseline --no-kernel-timers --backward --steps 10 --windows 2 > /dev/null 2>&1
  echo "== BEVMSDA_IMAGE_BATCH=$v"
  python - $GRAFT_REPO_ROOT/gpurun_out/ib_trace_$v/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    if "pack" in r["Name"] or "CatArrayBatchedCopy" in r["Name"]:
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.3f}")
print("all kernels total ms", tot / 1e6)
PY
done
