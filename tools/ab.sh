#!/bin/bash
# development A/B on the GPU box: k_primary / frame time of librt_hip.so (A) against librt_hip_b.so (B), interleaved
# usage: bash tools/ab.sh [k0_time args]
for i in 1 2; do
  echo -n "A: "; timeout -k 10 120 python3 -u tools/k0_time.py "$@" 2>&1 | grep "^mode" || exit 1
  echo -n "B: "; RT_HIP_LIB=raytracer_amd/librt_hip_b.so timeout -k 10 120 python3 -u tools/k0_time.py "$@" 2>&1 | grep "^mode" || exit 1
done
