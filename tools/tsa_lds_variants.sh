set -e
python tools/tsa_lds_kernel_ab.py
for v in ${VARIANTS:-d1 d2 d3}; do
  BEVMSDA_LIBRARY=$PWD/bevformer_amd/lib/libbevmsda_tsalds_$v.so python tools/tsa_lds_kernel_ab.py
done
