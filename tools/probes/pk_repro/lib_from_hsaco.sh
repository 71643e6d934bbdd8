#!/bin/bash
# libbevmsda_<name>.so whose sampling-backward unit carries the DEVICE code of tools/probes/pk_repro/hsaco/<name>.hsaco
# (the host half of the unit is compiled from source, the code object bundled in; every other object from the default build)
#   bash tools/probes/pk_repro/lib_from_hsaco.sh inplace_tmp
set -e
name=$1
root=$(cd "$(dirname "$0")/../../.." && pwd)
L=/opt/rocm/lib/llvm/bin
h=$root/tools/probes/pk_repro/hsaco
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
   -input=/dev/null -input=$h/$name.hsaco -output=$h/$name.hipfb
cd $root/bevformer_amd/csrc
obj=../lib/obj/bevmsda_capi_backward_$name.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed --cuda-host-only \
   -Xclang -fcuda-include-gpubinary -Xclang $h/$name.hipfb -c bevmsda_capi_backward.hip -o $obj
objs=$(ls ../lib/obj/*.o | grep -v "bevmsda_capi_backward" ; echo $obj)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libbevmsda_$name.so $objs
echo built bevformer_amd/lib/libbevmsda_$name.so
