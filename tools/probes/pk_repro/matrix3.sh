# third visit: load STEPS instead of steady load — idle gaps before every batch (and a GEMM in front), with a second process doing the same
cd $GRAFT_REPO_ROOT
H="python tools/probes/pk_repro/hunt.py"
$H --hsaco slp --contender self --idle-ms 20 --batch 8 --seconds 25
$H --hsaco slp --contender self --idle-ms 50 --batch 4 --gemm-first --seconds 25
$H --hsaco slp --contender self --idle-ms 5 --batch 4 --gemm-first --seconds 25
$H --hsaco slp --contender matmul --idle-ms 20 --batch 8 --seconds 25
$H --hsaco slp --contender none --idle-ms 20 --batch 8 --gemm-first --seconds 20
