# A/B on one box: the library built with the SLP vectorizer (packed fp32 math in the sampling backward too: tools/build_variant.sh slp bevmsda_capi_backward.hip -> bevformer_amd/lib/libbevmsda_slp.so) against the
# default build without it, interleaved: forward step and training step of the base shape set.
for rep in 1 2; do
  for v in noslp slp; do
    if [ $v = slp ]; then export BEVMSDA_LIBRARY=$PWD/bevformer_amd/lib/libbevmsda_slp.so; else unset BEVMSDA_LIBRARY; fi
    python bench.py --no-variants --no-cpu-baseline --steps 20 --windows 5 > gpurun_out/ab_${v}_fwd_$rep.json 2>/dev/null
    python bench.py --no-variants --no-cpu-baseline --backward --steps 10 --windows 3 > gpurun_out/ab_${v}_bwd_$rep.json 2>/dev/null
    echo "== $v rep $rep"; python tools/bench_digest.py gpurun_out/ab_${v}_fwd_$rep.json | grep -E "^value|kernel|gemms total|    "; python tools/bench_digest.py gpurun_out/ab_${v}_bwd_$rep.json | grep -E "^value"
  done
done
