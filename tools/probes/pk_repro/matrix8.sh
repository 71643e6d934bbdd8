# eighth visit: which op_sel pattern (B: a low lane reads a high half / C: a high lane reads a low half), and the source-level fix
cd $GRAFT_REPO_ROOT
F="python tools/probes/pk_repro/flow_hunt.py --contender self"
for v in sc_selB sc_selC; do
  BEVMSDA_LIBRARY=$PWD/bevformer_amd/lib/libbevmsda_$v.so timeout 200 $F --passes 4000 2>&1 | grep -v amdgpu.ids | cut -c1-420
done
timeout 300 $F --passes 10000 2>&1 | grep -v amdgpu.ids | cut -c1-420
for i in 1 2 3 4 5 6; do timeout 250 python tools/ddp_diag.py --tries 64 2>&1 | grep -E "tries with|Error:"; done
