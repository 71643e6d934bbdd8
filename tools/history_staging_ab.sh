# the history queue (4 frames / step): pinned staging ring of the per-frame pose copies (BEVMSDA_STAGING_SLOTS, 0 = blocking copies)
# x hoisted camera-value projection on a second stream (BEVMSDA_OVERLAP); prints ms per step and the host's time to ISSUE a step
run() { BEVMSDA_STAGING_SLOTS=$1 BEVMSDA_OVERLAP=$2 BEVMSDA_QUEUE_OVERLAP=$2 python bench.py --no-cpu-baseline --no-variants --steps 20 --windows 5 ${@:3} 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('slots=$1 overlap=$2 [${*:3}] ms_per_step %.4f host issue %.3f' % (l['ms_per_step'], l.get('host_issue_ms_per_step') or -1), (l.get('parity') or {}).get('ok'))"; }
for r in 1 2; do
run 8 1 --queue 4; run 0 1 --queue 4; run 8 0 --queue 4; run 0 0 --queue 4
done
run 8 1; run 8 0
run 8 1 --queue 4 --gemm bf16 --value-storage bf16; run 8 0 --queue 4 --gemm bf16 --value-storage bf16
