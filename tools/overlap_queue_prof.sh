# kernel statistics of the history-queue step with the hoisted camera-value projection on one stream / on a second stream
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/prof_$v
  BEVMSDA_OVERLAP=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python $root/bench.py --queue 4 --no-cpu-baseline --no-variants --no-kernel-timers --steps 5 --warmup 2 --windows 2 > /tmp/prof_$v.log 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== overlap=$v ($f)"; tail -2 /tmp/prof_$v.log | cut -c1-200
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:14]:
    print("%8.1f us x %5s  %6.2f ms  %s" % (float(r["AverageNs"])/1e3, r["Calls"], float(r["TotalDurationNs"])/1e6, r["Name"][:90]))
PY
done
