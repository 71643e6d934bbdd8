#!/bin/bash
# One library variant for an A/B: tools/build_variant.sh NAME SOURCE.hip [-DFOO=1 ...]  ->  bevformer_amd/lib/libbevmsda_NAME.so
# (SOURCE recompiled with the extra flags, every other object taken from the default build; select it with BEVMSDA_LIBRARY=...)
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root/bevformer_amd/csrc"
obj=../lib/obj/${src%.hip}_$name.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed "$@" -c $src -o $obj
# every default object except SOURCE's own (default objects are exactly the csrc/*.hip names; other variants' objects are skipped)
objs=$(for f in *.hip; do [ "$f" = "$src" ] || echo ../lib/obj/${f%.hip}.o; done; echo $obj)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libbevmsda_$name.so $objs
echo built bevformer_amd/lib/libbevmsda_$name.so
