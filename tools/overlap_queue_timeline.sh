# timeline (kernel start / end, queue) of one replayed history frame of the 4-frame queue: overlap on / off
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/tl_$v
  BEVMSDA_OVERLAP=$v timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$v -- python $root/bench.py ${TL_ARGS:---queue 4} --no-cpu-baseline --no-variants --no-kernel-timers --steps 3 --warmup 2 --windows 1 > /tmp/tl_$v.log 2>&1
  f=$(find /tmp/tl_$v -name "*kernel_trace.csv" | head -1)
  echo "== overlap=$v ($f)"
  python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows: r["s"]=int(r["Start_Timestamp"]); r["e"]=int(r["End_Timestamp"])
rows.sort(key=lambda r:r["s"])
# the last frame: walk back from the end to the 2nd-last flatten_feats launch group
idx=[i for i,r in enumerate(rows) if "flatten_feats" in r["Kernel_Name"] or "plan_project" in r["Kernel_Name"]]
# frames start at a gap: find starts of the last two frames by the first kernel after an idle gap or by name
starts=[i for i,r in enumerate(rows) if "plan_project" in r["Kernel_Name"]]
print("frames seen", len(starts), "columns", list(rows[0].keys())[:12])
# per frame: span from one plan_project to the next, the busy union of kernel intervals, idle gaps > 15 us
for k in range(max(0,len(starts)-13), len(starts)-1):
    fa, fb = starts[k], starts[k+1]
    seg = rows[fa:fb]
    span = (rows[fb]["s"]-rows[fa]["s"])/1e3
    busy = 0; cur_e = seg[0]["s"]; gaps=[]
    for r in seg + [rows[fb]]:
        if r["s"] > cur_e:
            if r["s"]-cur_e > 15000: gaps.append((round((cur_e-seg[0]["s"])/1e3), round((r["s"]-cur_e)/1e3,1)))
        if r is not rows[fb]:
            busy += max(0, r["e"]-max(cur_e, r["s"])); cur_e = max(cur_e, r["e"])
    hist = any("linear_panel_kernel<3, 2, 2, 4" in r["Kernel_Name"] for r in seg)
    print("frame %2d span %8.1f us busy %8.1f  n %3d  hist-panel %s  gaps(at,len) %s" % (k, span, busy/1e3, len(seg), hist, gaps[:6]))
if __import__("os").environ.get("TL_BRIEF"): sys.exit(0)
a=starts[-2]; b=starts[-1]
# include kernels before plan_project that belong to the frame (prologue): back up to 12 launches
t0=rows[a]["s"]
for r in rows[max(0,a-14):b]:
    print("%9.1f %9.1f  q%-3s %7.1f us  %s" % ((r["s"]-t0)/1e3,(r["e"]-t0)/1e3,r.get("Queue_Id","?"),(r["e"]-r["s"])/1e3,r["Kernel_Name"][9:80]))
PY
done
