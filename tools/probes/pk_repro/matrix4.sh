# fourth visit: the round-5 scenario (tools/ddp_diag.py, 2 ranks x 64 tries per run) with the packed build (control, rebuilt
# through the same assemble-and-bundle path) against the variant whose 28 in-place cross-half packed instructions read a copy
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  for v in slp inplace_tmp; do
    echo "== $v run $i"
    BEVMSDA_LIBRARY=$PWD/bevformer_amd/lib/libbevmsda_$v.so timeout 250 python tools/ddp_diag.py --tries 64 2>&1 | grep -E "tries with|Error|error"
  done
done
H="python tools/probes/pk_repro/hunt.py"
$H --hsaco slp --contender self --seconds 15 --rows 46000 --shapes 116x200,58x100,29x50,15x25 --batch 4
