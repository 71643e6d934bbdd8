run() { for i in 1 2 3 4; do env "$@" python -m pytest tests/test_ddp_gpu.py -q 2>&1 | grep -E 'passed|failed' ; done; }
echo "== default"; run A=1
echo "== no arena"; run BEVMSDA_GRAD_ARENA=0
echo "== no flatten"; run BEVMSDA_FLATTEN_PARAMS=0
echo "== no weight views"; run BEVMSDA_WEIGHT_VIEWS=0
echo "== sync before passes"; run DDP_TEST_SYNC=1
echo "== no train chain"; run BEVMSDA_TRAIN_CHAIN=0
