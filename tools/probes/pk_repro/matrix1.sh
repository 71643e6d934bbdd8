# first visit: does the operator-level loop reproduce the defect at all, and under which contender
cd $GRAFT_REPO_ROOT
H="python tools/probes/pk_repro/hunt.py"
for c in self matmul tiny none thread idle; do $H --hsaco slp --contender $c --seconds 12; done
$H --hsaco noslp --contender self --seconds 12
$H --hsaco slp --contender self --seconds 12 --points 4
$H --hsaco slp --contender self --seconds 12 --rows 46000 --shapes 116x200,58x100,29x50,15x25
