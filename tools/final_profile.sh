#!/bin/bash
# the round's committed evidence: the DEFAULT bench command under rocprofv3 --kernel-trace --stats
# usage (GPU box, repo root): bash tools/final_profile.sh <tag>
tag=${1:-final}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py > gpurun_out/prof_$tag.log 2>&1 || exit 1
grep "^{" gpurun_out/prof_$tag.log > gpurun_out/bench_under_prof_$tag.json
python3 tools/frame_timeline.py gpurun_out/prof_$tag
