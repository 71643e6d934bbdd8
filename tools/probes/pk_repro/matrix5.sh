# fifth visit: which kernel's packed math matters in the round-5 scenario (2 ranks x 64 tries per run, 6 runs per variant, interleaved)
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  for v in gl_noslp sort_noslp glsort_noslp noslp nop_all wait_all slp; do
    echo "== $v run $i"
    BEVMSDA_LIBRARY=$PWD/bevformer_amd/lib/libbevmsda_$v.so timeout 250 python tools/ddp_diag.py --tries 64 2>&1 | grep -E "tries with|Error:"
  done
done
