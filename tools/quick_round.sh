#!/bin/bash
# development: bench (100 steps) + kernel timeline of a short profiled run, no test suite
tag=${1:-q}
timeout -k 10 200 python3 -u bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('step_ms', round(d['ms_per_step'],4), 'k0_ms', round(d['roofline']['avg_launch_ms'],4), 'Mrays/s', round(d['value']))" || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1 || exit 1
python3 tools/frame_timeline.py gpurun_out/prof_$tag
