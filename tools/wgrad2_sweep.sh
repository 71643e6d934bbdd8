# the 256 x 128 weight-gradient shape at several workgroup targets (one 512-thread workgroup per CU: 256 resident)
cd $GRAFT_REPO_ROOT
for wg in 192 224 256 384 512; do
  echo "== variant 2, workgroups $wg"
  BEVMSDA_WGRAD_VARIANT=2 BEVMSDA_WGRAD_WGS=$wg python bench.py --no-cpu-baseline --no-variants --backward --graph off --steps 3 --warmup 2 --windows 1 2>/dev/null | python /tmp/wgrad_ab_digest.py gemms
done
echo "== variant 0 (default)"
BEVMSDA_WGRAD_VARIANT=0 python bench.py --no-cpu-baseline --no-variants --backward --graph off --steps 3 --warmup 2 --windows 1 2>/dev/null | python /tmp/wgrad_ab_digest.py gemms
