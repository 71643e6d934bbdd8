"""Pivot gpurun_out/gbench.json: rows = op/mode, columns = launch variants, cells = median us."""
import collections
import json
import sys

rows = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gbench.json"))
variants = sorted({r["variant"] for r in rows if r["variant"] is not None})
table = collections.OrderedDict()
errs = collections.defaultdict(float)
for r in rows:
    table.setdefault((r["op"], r["mode"]), {})[r["variant"]] = r["us"]
    errs[(r["mode"], r["variant"])] = max(errs[(r["mode"], r["variant"])], r["max_err_vs_native"])
print("op/mode".ljust(28) + "".join(f"v{v}".rjust(8) for v in ["dflt"] + variants))
tot = collections.defaultdict(float)
for (op, mode), d in table.items():
    print(f"{op}/{mode}".ljust(28) + "".join((f"{d[v]:8.1f}" if v in d else "       -") for v in [None] + variants))
    for v, us in d.items():
        tot[(mode, v)] += us
for mode in ("native", "split", "bf16"):
    line = "".join((f"{tot[(mode, v)]:8.1f}" if (mode, v) in tot else "       -") for v in [None] + variants)
    print(f"SUM/{mode}".ljust(28) + line)
print("max err vs native:", {f"{m}/v{v}": f"{e:.1e}" for (m, v), e in errs.items() if m != "native"})
