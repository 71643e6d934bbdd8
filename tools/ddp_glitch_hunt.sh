# Glitch hunt (profiles/r5/r5_ddp_forensics.txt): tools/ddp_diag.py N times (two ranks sharing the GPU, 64 tries of one DDP pass +
# three direct passes each) under the environment given as arguments; prints the per-rank count of disagreeing passes.
#   N=6 bash tools/ddp_glitch_hunt.sh A=1                         # the default build
#   BEVMSDA_LIBRARY=$PWD/bevformer_amd/lib/libbevmsda_slp.so ...        # packed fp32 math in the sampling backward too (tools/build_variant.sh slp bevmsda_capi_backward.hip)
N=${N:-6}
echo "=== $*"
for i in $(seq $N); do env "$@" timeout 250 python tools/ddp_diag.py --tries 64 2>&1 | grep -E "tries with"; done
