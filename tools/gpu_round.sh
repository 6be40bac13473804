#!/bin/bash
# one GPU round: parity suite, then (only if green) the bench and a kernel-trace profile of it
# usage (on the GPU box, from the repo root): bash tools/gpu_round.sh <tag>
tag=${1:-x}
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gpu_tests_$tag.log 2>&1
rc=$?
tail -5 gpurun_out/gpu_tests_$tag.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python3 -u bench.py --steps 30 --warmup 5 2>&1 | tail -1 | tee gpurun_out/bench_$tag.json || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1 || exit 1
python3 tools/frame_timeline.py gpurun_out/prof_$tag
