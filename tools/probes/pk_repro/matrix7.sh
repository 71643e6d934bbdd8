# seventh visit: which CLASS of packed instruction inside the gradloc kernels carries the defect (flow_hunt, contender = self)
cd $GRAFT_REPO_ROOT
for v in sc_all sc_add sc_addsel sc_mulfma sc_sel sc_nosel; do
  python tools/probes/pk_repro/hunt.py --hsaco $v --contender none --seconds 1 2>&1 | grep -v amdgpu.ids | cut -c1-330
  BEVMSDA_LIBRARY=$PWD/bevformer_amd/lib/libbevmsda_$v.so timeout 200 python tools/probes/pk_repro/flow_hunt.py --passes 4000 --contender self 2>&1 | grep -v amdgpu.ids | cut -c1-420
done
BEVMSDA_LIBRARY=$PWD/bevformer_amd/lib/libbevmsda_slp.so timeout 200 python tools/probes/pk_repro/flow_hunt.py --passes 4000 --contender self 2>&1 | grep -v amdgpu.ids | cut -c1-300
cat /sys/module/amdgpu/parameters/{cwsr_enable,sched_policy,hws_max_conc_proc,halt_if_hws_hang,noretry,mes} 2>&1 | tr '\n' ' '
