# Interleaved A/B on one box: SCA sampling over the device-side row count as hint-sized launch + tail launch (0) vs ONE
# capacity-sized launch (1).
for rep in 1 2; do
  for v in 0 1; do
    BEVMSDA_FUSED_CAPACITY=$v python bench.py --no-variants --no-cpu-baseline --steps 20 --windows 5 > gpurun_out/ab_cap_${v}_$rep.json 2>/dev/null
    echo "== BEVMSDA_FUSED_CAPACITY=$v rep $rep: $(python tools/bench_digest.py gpurun_out/ab_cap_${v}_$rep.json | grep -E '^value|sca_fwd|parity' | cut -c1-150 | tr '\n' ' ')"
  done
done
