#!/bin/bash
# development: build librt_hip with AddressSanitizer + UBSan on the HOST code only (device code as shipped) and the
# driver tools/host_asan_driver.c against it, into tools/asan_out/ (CPU, here); run tools/asan_out/driver on the GPU box
# with ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=0:protect_shadow_gap=0 (take tools/asan_out out of
# .gpurunignore for that call: it is listed there so that ordinary calls do not upload 12 MB of sanitizer builds).
set -e
cd "$(dirname "$0")/.."
S=raytracer_amd/csrc; O=tools/asan_out; mkdir -p $O
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer -g"
HF="--offload-arch=gfx950 -O2 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Iinclude -I$S"
for f in rt_kernels rt_api rt_tone; do /opt/rocm/bin/hipcc $HF $SAN -c $S/$f.hip -o $O/$f.o; done
/opt/rocm/bin/hipcc $HF -DRT_FAST=1 -ffp-contract=fast $SAN -c $S/rt_kernels.hip -o $O/rt_kernels_fast.o
make -s -C $S build/rt_jit_src.inc
/opt/rocm/bin/hipcc $HF $SAN -I$S/build -x hip -c $S/rt_jit.cpp -o $O/rt_jit.o
gcc -O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -Iinclude -I$S -c $S/rt_host.c -o $O/rt_host.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fsanitize=undefined -fno-gpu-sanitize -o $O/librt_hip.so $O/*.o -L/opt/rocm/lib -lhiprtc -lm
/opt/rocm/bin/hipcc -x c -O1 -g -fsanitize=address -fsanitize=undefined -fno-gpu-sanitize -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
    tools/host_asan_driver.c -o $O/driver -L$O -lrt_hip -Wl,-rpath,'$ORIGIN' -L/opt/rocm/lib -lamdhip64 -lm
echo built $O/driver
