# sixth visit: what the second process has to be doing (one hunter process, no DDP, no gloo; packed build)
cd $GRAFT_REPO_ROOT
export BEVMSDA_LIBRARY=$PWD/bevformer_amd/lib/libbevmsda_slp.so
F="python tools/probes/pk_repro/flow_hunt.py --passes 5000"
for c in self none matmul tiny idle thread; do timeout 300 $F --contender $c 2>&1 | grep -v amdgpu.ids | cut -c1-900; done
unset BEVMSDA_LIBRARY
timeout 300 $F --contender self 2>&1 | grep -v amdgpu.ids | cut -c1-600
