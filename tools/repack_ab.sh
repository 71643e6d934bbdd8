# Interleaved A/B on one box: inference graph with the cached weight images (default) vs re-packing them in every replay
# (BEVMSDA_GRAPH_REPACK=1, round 4's behaviour).
for rep in 1 2; do
  for v in 0 1; do
    BEVMSDA_GRAPH_REPACK=$v python bench.py --no-variants --no-cpu-baseline --steps 20 --windows 5 > gpurun_out/ab_repack_${v}_$rep.json 2>/dev/null
    echo "== BEVMSDA_GRAPH_REPACK=$v rep $rep: $(python tools/bench_digest.py gpurun_out/ab_repack_${v}_$rep.json | head -2 | tr '\n' ' ')"
  done
done
