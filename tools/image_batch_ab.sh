# A/B on one box, interleaved: the training step with the one-launch weight-image rebuild (default) against image-by-image
# packing (BEVMSDA_IMAGE_BATCH=0): base fp32 and small4 bf16 (BASELINE configs[2]).
for rep in 1 2; do
  for v in 1 0; do
    export BEVMSDA_IMAGE_BATCH=$v
    python bench.py --no-variants --no-cpu-baseline --backward --steps 10 --windows 3 > gpurun_out/ib_${v}_base_$rep.json 2>/dev/null
    python bench.py --no-variants --no-cpu-baseline --backward --workload small4 --gemm bf16 --value-storage bf16 --steps 10 --windows 3 > gpurun_out/ib_${v}_small4_$rep.json 2>/dev/null
    echo "== image batch $v rep $rep: base $(tail -n 1 gpurun_out/ib_${v}_base_$rep.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["launch_mode"])')  small4 bf16 $(tail -n 1 gpurun_out/ib_${v}_small4_$rep.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["ms_per_step"],3), d["launch_mode"])')"
  done
done
