#!/bin/bash
# A/B of the weight-gradient kernels inside the forward + backward step (GPU box): VARIANTS="0 2" tools/wgrad_ab.sh
# (0 = bf16 planes + transposing LDS reads, 1 = gathered fragments, 2 = 256 x 128 tiles on 8 wavefronts with 2 LDS stages)
cat > /tmp/wgrad_ab_digest.py <<'PY'
import json, sys
lines = [l for l in sys.stdin if l.startswith("{")]
full = [json.loads(l)["bench_detail"] for l in lines if l.startswith('{"bench_detail"')]
d = full[-1] if full else json.loads(lines[-1])
if sys.argv[1] == "step":
    print("   graph replay: ms_per_step %.3f" % d["ms_per_step"])
else:
    g, tot = d["gemms"]["per_tag"], 0
    for k, v in sorted(g.items()):
        if "_dw" in k:
            print("   %-34s %7.1f us x%d" % (k, v["avg_us"], v["launches"]))
            tot += v["avg_us"] * v["launches"] / 3
    print("   weight gradients %.0f us per step" % tot)
PY
for round in 1 2; do
for v in ${VARIANTS:-1 0}; do
  echo "== BEVMSDA_WGRAD_VARIANT=$v ${WORKLOAD_ARGS:-}"
  BEVMSDA_WGRAD_VARIANT=$v python bench.py --no-cpu-baseline --no-variants --backward --steps 5 --warmup 2 --windows 3 ${WORKLOAD_ARGS:-} 2>/dev/null | python /tmp/wgrad_ab_digest.py step
  BEVMSDA_WGRAD_VARIANT=$v python bench.py --no-cpu-baseline --no-variants --backward --graph off --steps 3 --warmup 2 --windows 1 ${WORKLOAD_ARGS:-} 2>/dev/null | python /tmp/wgrad_ab_digest.py gemms
done; done
