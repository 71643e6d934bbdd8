"""Print the headline fields of a bench.py JSON line (last line of the file)."""
import json
import sys

line = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")]
if not line:
    print("no JSON line in", sys.argv[1])
    sys.exit(0)
# round 6: the full record is the {"bench_detail": ...} line printed BEFORE the compact final record
full = [json.loads(l) for l in line if l.startswith('{"bench_detail"')]
d = full[-1]["bench_detail"] if full else json.loads(line[-1])
print(f"value {d['value']:.4g} {d['unit']}  ms/step {d['ms_per_step']:.3f}  mode {d.get('launch_mode')}  dtype {d.get('dtype')}")
print("windows", d.get("windows", {}).get("ms_per_step"), "geometry_ms", d.get("geometry_ms"))
r = d.get("roofline", {})
print(f"roofline: {r.get('avg_us'):.1f} us  {r.get('achieved'):.0f} GB/s  frac {r.get('frac'):.3f}  rows {d['config'].get('sca_rows_per_frame')} order {d['config'].get('sca_row_order')}")
for k, v in (d.get("kernels") or {}).items():
    print(f"  kernel {k}: {v['avg_us']:.1f} us x{v['launches']}")
g = d.get("gemms")
if g:
    print(f"  gemms total {g['total_us_per_step']:.0f} us/step  {g['TFLOPs']:.0f} TFLOP/s  mfma frac {g.get('frac_of_bf16_mfma_peak_2500')}")
    for k, v in g["per_tag"].items():
        print(f"    {k}: {v['avg_us']:.1f} us x{v['launches']}  {v['TFLOPs']:.0f} TF  {v['alg_GBs']:.0f} GB/s")
if "parity" in d:
    print("parity", d["parity"])
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print(f"cpu {c['value']:.0f} q/s on {c['cores']} threads, runs {c.get('runs_s')}; vs_cpu {d.get('vs_cpu_baseline'):.0f}x")
for k, v in (d.get("variants") or {}).items():
    print(f"  variant {k}: {v['ms_per_step']:.3f} ms (min {v.get('ms_per_step_min', float('nan')):.3f}) {v['launch_mode']} parity {v.get('parity')}")
for key in ("multi_gpu_model", "multi_gpu_model_bf16"):
    m = d.get(key)
    if m:
        print(f"  {key}: t1 {m['t1_ms']:.3f} ms")
        for g in ("2", "4", "8"):
            e = m[g]
            print(f"    G={g} ({e['layout']}): per-rank max {max(e['per_rank_ms']):.3f} ms + all-gather {e['all_gather_model_us']:.0f} us -> "
                  f"efficiency {e['efficiency']:.3f} (Amdahl bound {e['amdahl_bound_efficiency']})")
