#!/bin/bash
# development: build a variant of the library into tools/abl/librt_<TAG>.so with extra -D flags
# usage: tools/build_abl.sh TAG -DRT_PHASE_CLOCKS ...
set -e
tag=$1; shift
cd "$(dirname "$0")/../raytracer_amd/csrc"
mkdir -p build/abl ../../tools/abl
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -I../../include -I."
/opt/rocm/bin/hipcc $F "$@" -c rt_kernels.hip -o build/abl/k_$tag.o
make -s build/rt_api.o build/rt_tone.o build/rt_host.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/abl/librt_$tag.so build/abl/k_$tag.o build/rt_api.o build/rt_tone.o build/rt_host.o -lm
