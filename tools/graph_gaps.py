"""Where a replayed step's time goes between kernels: reads a rocprofv3 ``--kernel-trace`` CSV and prints, over the
last ``--last`` kernel dispatches, the sum of kernel durations, the sum of the idle gaps between consecutive kernels
(end -> next start, same device) and the largest gaps with the kernels on either side.

    gpurun -- 'cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gaps -- python $GRAFT_REPO_ROOT/bench.py --workload tiny --no-variants --no-cpu-baseline --steps 50; python $GRAFT_REPO_ROOT/tools/graph_gaps.py $GRAFT_REPO_ROOT/gpurun_out/gaps'"""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 2000
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print("no *kernel_trace.csv under", root)
        return
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[-last:]
    busy = sum(e - s for s, e, _ in rows)
    gaps = []
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        gaps.append((s1 - e0, n0, n1))
    span = rows[-1][1] - rows[0][0]
    small = [g for g, _, _ in gaps if 0 <= g < 20000]
    print(f"{len(rows)} dispatches over {span / 1e3:.1f} us: kernels {busy / 1e3:.1f} us ({busy / span:.1%}), "
          f"gaps < 20 us: {sum(small) / 1e3:.1f} us in {len(small)} gaps (mean {sum(small) / max(1, len(small)) / 1e3:.2f} us), "
          f"overlapping pairs {sum(1 for g, _, _ in gaps if g < 0)}")
    import collections
    per = collections.defaultdict(lambda: [0, 0, 0])
    for (s, e, n), g in zip(rows[1:], gaps):
        k = n.split("(")[0][:70]
        per[k][0] += 1
        per[k][1] += e - s
        per[k][2] += max(0, min(g[0], 20000))
    if "--step" in sys.argv:
        # one whole step from the middle of the trace: the dispatches between two launches of the marker kernel
        marker = sys.argv[sys.argv.index("--step") + 1]
        idx = [i for i, r in enumerate(rows) if marker in r[2]]
        if len(idx) > 4:
            a, b = idx[len(idx) // 2], idx[len(idx) // 2 + 1]
            t0 = rows[a][0]
            print(f"one step ({b - a} dispatches, {(rows[b][0] - t0) / 1e3:.1f} us start to start):")
            for i in range(a, b):
                s_, e_, n_ = rows[i]
                gap = rows[i][0] - rows[i - 1][1]
                print(f"  +{(s_ - t0) / 1e3:8.1f} us  gap {gap / 1e3:6.2f}  dur {(e_ - s_) / 1e3:7.2f}  {n_.split('(')[0][:90]}")
    print("kernel: launches, mean duration us, mean gap in front of it us")
    for k, (c, d, g) in sorted(per.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"  {k:70s} {c:5d} {d / c / 1e3:8.2f} {g / c / 1e3:7.2f}")


if __name__ == "__main__":
    main()
