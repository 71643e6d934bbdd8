"""Condense a tools/prof.sh output directory into two small files that are
committed under profiles/: kernel_stats.csv (rocprofv3 --stats, names shortened)
and pmc.json (per-dispatch counter averages of the sampling kernels)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(n):
    if n.startswith("Cijk"):
        m = re.search(r"MT(\d+x\d+x\d+)", n)
        return "hipblaslt_gemm_f32 MT" + (m.group(1) if m else "?")
    return n.replace("void ", "").replace("at::native::", "")[:110]


def main(out):
    for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
        agg = collections.OrderedDict()
        for r in rows:
            a = agg.setdefault(short(r["Name"]), [0, 0.0])
            a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
        with open(os.path.join(out, "kernel_stats.csv"), "w") as w:
            w.write("percent,calls,avg_us,total_ms,kernel\n")
            for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                w.write(f"{t / tot * 100:.2f},{c},{t / c / 1e3:.1f},{t / 1e6:.3f},\"{k}\"\n")
        print(open(os.path.join(out, "kernel_stats.csv")).read()[:3000])
    pmc = {}
    for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if os.environ.get("PROF_BY_GRID") and r.get("Grid_Size"):
                name += f" grid={r['Grid_Size']}"
            k = (name, r["Counter_Name"])
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
        for (k, c), (v, n) in agg.items():
            pmc.setdefault(k, {})[c] = dict(per_dispatch=v / n, dispatches=n)
    for k, d in pmc.items():
        # HBM traffic per launch, MI355X_MICROARCH.md §HBM: FETCH_SIZE (KB) counts 128-B requests
        # as 64 B for wide coalesced reads -> doubled; WRITE_SIZE (KB) taken as is
        if "FETCH_SIZE" in d:
            d["hbm_read_bytes_corrected"] = d["FETCH_SIZE"]["per_dispatch"] * 1024 * 2
        if "WRITE_SIZE" in d:
            d["hbm_write_bytes"] = d["WRITE_SIZE"]["per_dispatch"] * 1024
    json.dump(pmc, open(os.path.join(out, "pmc.json"), "w"), indent=1)
    print(json.dumps(pmc, indent=1)[:4000])


if __name__ == "__main__":
    main(sys.argv[1])
