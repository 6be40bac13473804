#!/bin/bash
# development: alone-on-the-GPU kernel times of the big scenes for the libraries given (A/B builds).  usage: tools/ab_big.sh "c4 c5s" librt_hip.so librt_hip_b.so
cfgs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  for c in $cfgs; do
    RT_HIP_LIB=raytracer_amd/$lib timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ab_${lib}_$c -- python3 bench.py --config $c --numerics exact --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot --no-generic --shared-stream --inflight 1 --tail-stream off > gpurun_out/ab_${lib}_$c.log 2>&1 || { tail -5 gpurun_out/ab_${lib}_$c.log; exit 1; }
    echo "== $lib $c"; python3 tools/big_trace.py gpurun_out/ab_${lib}_$c | grep -E "k_trace|k_hits|k_shadefold|k_resolve|sum"
  done
done
