# the hoisted camera-value projection on a second stream (BEVMSDA_OVERLAP) by configuration, interleaved on one box
run() { BEVMSDA_OVERLAP=$1 python bench.py --no-cpu-baseline --no-variants --steps 20 --windows 5 "${@:3}" 2>/dev/null | tail -1 | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print('$2 overlap=$1 ms_per_step %.4f' % l['ms_per_step'], (l.get('parity') or {}).get('ok'))"; }
for r in 1 2; do
  for v in 1 0; do
    for c in ${CONFIGS:-base queue4 tiny small4 rank0of8 rank0of2 first_frame}; do
      case $c in
        base) run $v base ;;
        queue4) run $v queue4 --queue 4 ;;
        queue4_bf16) run $v queue4_bf16 --queue 4 --gemm bf16 --value-storage bf16 ;;
        tiny) run $v tiny --workload tiny ;;
        small4) run $v small4 --workload small4 ;;
        rank0of8) run $v rank0of8 --simulate-rank 0,8 ;;
        rank0of2) run $v rank0of2 --simulate-rank 0,2 ;;
        first_frame) run $v first_frame --first-frame ;;
      esac
    done
  done
done
