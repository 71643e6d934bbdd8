set -e
python tools/tsa_lds_kernel_ab.py
for v in "d1 -DBEVMSDA_TSA_LDS_DIAG=1" "d2 -DBEVMSDA_TSA_LDS_DIAG=2" "d3 -DBEVMSDA_TSA_LDS_DIAG=3" "d4 -DBEVMSDA_TSA_LDS_DIAG=4" "d7 -DBEVMSDA_TSA_LDS_DIAG=7" "ty16 -DBEVMSDA_TSA_LDS_TY=16"; do
  set -- $v
  BEVMSDA_LIBRARY=$PWD/bevformer_amd/lib/libbevmsda_tsalds_$1.so python tools/tsa_lds_kernel_ab.py
done
