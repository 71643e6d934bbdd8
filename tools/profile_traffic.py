"""HBM traffic and kernel statistics of a bench configuration, reproducibly:

    gpurun -- python tools/profile_traffic.py [--config base_fwd | small4_bwd_bf16 | base_bwd] [--tag r3]
    python tools/profile_traffic.py --install [--tag r3]          (here, after gpurun merged gpurun_out/)

On the GPU box it runs ``bench.py`` (eager launches, 3 steps) under rocprofv3: one ``--kernel-trace --stats`` pass and
SEPARATE ``--pmc`` passes (never combined with tracing domains; tools/prof.sh), condenses them (tools/prof_summary.py)
and derives the HBM bytes per launch of the sampling kernels exactly as MI355X_MICROARCH.md prescribes
(FETCH_SIZE [KB] x 1024 x 2 — gfx950 tallies the 128-byte requests of wide coalesced reads at 64 bytes — plus
WRITE_SIZE [KB] x 1024).  Everything lands in ``gpurun_out/profile_traffic/<config>/``; ``--install`` copies the
summaries to ``profiles/<tag>/`` and rewrites ``profiles/traffic.json`` (what ``bench.py`` quotes as
``roofline.traffic_from_profile``)."""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = {
    # BASELINE.json headline: bevformer_base forward, fp32 storage, split-bf16 GEMMs
    "base_fwd": dict(workload="base", bench=["--workload", "base"], passes="1 2 5 6"),
    # BASELINE.json configs[2]: "small ... fwd+bwd bf16 ... rocprof HBM/MFMA counters"
    "small4_bwd_bf16": dict(workload="small4", bench=["--workload", "small4", "--backward", "--value-storage", "bf16", "--gemm", "bf16"],
                            passes="1 2 5 6"),
    "base_bwd": dict(workload="base", bench=["--workload", "base", "--backward"], passes="1 2 5 6"),
}


def classify(name):
    """Kernel name -> short tag of the hand-written kernels whose traffic is recorded."""
    m = re.search(r"msda_fused_d32(?:_bf16x8)?_(?:head|dyn|)_?kernel<(.*?)>", name)
    if m and "dyn" not in name:
        a = [x.strip() for x in m.group(1).split(",")]
        nums = [x for x in a if x.isdigit()]
        if len(nums) >= 2:
            return {("8", "1"): "sca_fwd", ("4", "2"): "tsa_fwd", ("4", "1"): "tsa_fwd_first_frame"}.get((nums[0], nums[1]))
    if "msda_fused_d32_tsa_pipe_kernel" in name:      # round 6: TSA's resident-grid form (K = 2 queue entries, 4 points)
        return "tsa_fwd"
    if "msda_gradvalue_sort_kernel" in name:
        return "grad_value_sort:" + re.sub(r".*kernel", "", name)
    if "msda_gradloc_d32" in name:
        return "grad_loc_gather:" + re.sub(r".*kernel", "", name)
    if "linear_chain_kernel" in name:
        return "linear_chain:" + re.sub(r".*kernel", "", name)
    if "linear_panel_kernel" in name:
        return "linear_panel:" + re.sub(r".*kernel", "", name)
    if "linear_splitbf16_kernel" in name or "linear_pipe_kernel" in name:
        return "linear_first:" + re.sub(r".*kernel", "", name)
    if "wgrad_splitbf16_kernel" in name or "wgrad_multi_kernel" in name:
        return "linear_wgrad:" + re.sub(r".*kernel", "", name)
    if "wgrad_tr_multi_kernel" in name:
        return "linear_wgrad_tr:" + re.sub(r".*kernel", "", name)
    return None


def collect(config, out):
    c = CONFIGS[config]
    env = dict(os.environ, TRACE="1", PMC="1", PMC_ONLY=c["passes"], PASS_TIMEOUT="240")
    cmd = ["bash", os.path.join(ROOT, "tools", "prof.sh"), "traffic_" + config, sys.executable, os.path.join(ROOT, "bench.py"),
           "--no-cpu-baseline", "--no-variants", "--graph", "off", "--steps", "3", "--warmup", "1", "--windows", "1"] + c["bench"]
    log = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True)
    src = os.path.join(ROOT, "gpurun_out", "prof_traffic_" + config)
    os.makedirs(out, exist_ok=True)
    open(os.path.join(out, "prof.log"), "w").write(log.stdout[-20000:] + "\n--- stderr ---\n" + log.stderr[-5000:])
    for f in ("kernel_stats.csv", "pmc.json", "failed_passes.txt"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(out, f))
    derive(config, out)


def derive(config, out):
    """pmc.json of a collected configuration -> traffic.json (also ``--rederive`` here, e.g. after ``classify`` learnt a new kernel name)."""
    c = CONFIGS[config]
    pmc = json.load(open(os.path.join(out, "pmc.json")))
    kernels = {}
    for name, d in pmc.items():
        tag = classify(name)
        if tag is None:
            continue
        rd, wr = d.get("hbm_read_bytes_corrected"), d.get("hbm_write_bytes")
        e = dict(kernel=name, hbm_read_bytes=rd, hbm_write_bytes=wr,
                 hbm_bytes_per_launch=(rd + wr) if rd is not None and wr is not None else None,
                 dispatches=next(iter(d.values()))["dispatches"] if d else 0)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
            # matrix-pipe utilisation: busy cycles over (SIMDs x per-XCD active cycles); GRBM_GUI_ACTIVE sums the 8 XCDs
            e["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"]["per_dispatch"] / (1024 * d["GRBM_GUI_ACTIVE"]["per_dispatch"] / 8)
        for k in ("SQ_INSTS_MFMA", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "TCC_HIT_sum", "TCC_MISS_sum", "GRBM_GUI_ACTIVE"):
            if k in d:
                e[k] = d[k]["per_dispatch"]
        kernels[tag] = e
    res = dict(config=config, workload=c["workload"], bench_args=c["bench"], kernels=kernels,
               method="rocprofv3 --pmc in separate passes of `bench.py --graph off --steps 3 --warmup 1`; HBM bytes = "
                      "FETCH_SIZE(KB)*1024*2 + WRITE_SIZE(KB)*1024 (MI355X_MICROARCH.md, HBM section)")
    json.dump(res, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    print(json.dumps({k: (v["hbm_bytes_per_launch"], v.get("mfma_busy_frac")) for k, v in kernels.items()}, indent=1))
    print("kernel stats (top):")
    if os.path.exists(os.path.join(out, "kernel_stats.csv")):
        print("".join(open(os.path.join(out, "kernel_stats.csv")).readlines()[:16]))


def kernel_sources_sha():
    """Hash of EVERY source of the library (csrc/*.h, csrc/*.hip: kernels, their launchers and the selection logic in
    the C ABI units): ``bench.py`` quotes ``profiles/traffic.json`` only while it matches the tree — any change to the
    native code makes ``roofline.traffic`` null instead of silently stale."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "bevformer_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def install(tag):
    base = os.path.join(ROOT, "gpurun_out", "profile_traffic")
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    tj_path = os.path.join(ROOT, "profiles", "traffic.json")
    tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
    for config in sorted(os.listdir(base)) if os.path.isdir(base) else []:
        src = os.path.join(base, config)
        if not os.path.exists(os.path.join(src, "traffic.json")):
            continue
        for f in ("kernel_stats.csv", "pmc.json", "traffic.json"):
            if os.path.exists(os.path.join(src, f)):
                shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_traffic_{config}_{f}"))
        t = json.load(open(os.path.join(src, "traffic.json")))
        if config.endswith("_fwd"):
            entry = tj.setdefault(t["workload"], {})
            for k in ("sca_fwd", "tsa_fwd"):
                if k in t["kernels"] and t["kernels"][k]["hbm_bytes_per_launch"]:
                    entry[k] = t["kernels"][k]["hbm_bytes_per_launch"]
            tj["_source"] = f"profiles/{tag}/{tag}_traffic_{config}_pmc.json (python tools/profile_traffic.py --config {config})"
            tj["_comment"] = t["method"]
            # (the profile was collected on the snapshot gpurun sent: the tree at install time, if nothing was edited between)
            tj["_kernel_sources_sha"] = kernel_sources_sha()
        print("installed", config, "->", dst)
    json.dump(tj, open(tj_path, "w"), indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="base_fwd", choices=sorted(CONFIGS))
    ap.add_argument("--tag", default="r3")
    ap.add_argument("--install", action="store_true")
    ap.add_argument("--rederive", action="store_true", help="recompute traffic.json of every collected configuration from its pmc.json")
    a = ap.parse_args()
    if a.rederive:
        base = os.path.join(ROOT, "gpurun_out", "profile_traffic")
        for config in sorted(os.listdir(base)):
            if config in CONFIGS and os.path.exists(os.path.join(base, config, "pmc.json")):
                derive(config, os.path.join(base, config))
    if a.install:
        install(a.tag)
    elif a.rederive:
        pass
    else:
        collect(a.config, os.path.join(ROOT, "gpurun_out", "profile_traffic", a.config))


if __name__ == "__main__":
    main()
