#!/bin/bash
# development: big-scene parity tests, alone-on-the-GPU kernel times of c4 (and c5s), phase times of the counting build
tag=$1
timeout -k 10 600 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread -k "c4_sparse or c5_sparse or big_scene or grid_search or random_scenes" > gpurun_out/${tag}_tests.log 2>&1 || { tail -30 gpurun_out/${tag}_tests.log; exit 1; }
tail -1 gpurun_out/${tag}_tests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in ${2:-c4 c5s}; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_trace_$c -- python3 bench.py --config $c --numerics exact --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot --no-generic --shared-stream --inflight 1 --tail-stream off > gpurun_out/${tag}_trace_$c.log 2>&1 || { tail -5 gpurun_out/${tag}_trace_$c.log; exit 1; }
  echo "== $c"; python3 tools/big_trace.py gpurun_out/${tag}_trace_$c
done
if [ -f raytracer_amd/librt_hip_b.so ]; then
  JOBS_TIMES=1 RT_HIP_LIB=raytracer_amd/librt_hip_b.so timeout -k 10 200 python tools/jobs_count.py c4 | grep -v "sphere tests"
fi
