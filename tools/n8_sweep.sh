#!/bin/bash
# development: a rank of eight on config 3 (rehearsal), pipelining options
show() { python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('%-40s %.4f ms/frame  k_primary %.4f' % (sys.argv[1], d['ms_per_step'], d['roofline']['avg_launch_ms']))" "$1"; }
export RT_BENCH_REHEARSE_WORLD=8
B="timeout -k 10 120 python3 bench.py --config c3 --no-cpu-baseline --no-generic"
$B 2>/dev/null | show "default" || exit 1
$B --overlap 2>/dev/null | show "overlap 3" || exit 1
$B --overlap --inflight 4 2>/dev/null | show "overlap 4" || exit 1
$B --overlap --inflight 6 2>/dev/null | show "overlap 6" || exit 1
$B --render-priority normal 2>/dev/null | show "equal priorities" || exit 1
$B --render-priority normal --inflight 5 2>/dev/null | show "equal priorities, 5 in flight" || exit 1
$B --gather-stream same 2>/dev/null | show "gather on the render stream" || exit 1
$B --overlap --launch-plan pairs --inflight 4 2>/dev/null | show "overlap 4, pairs" || exit 1
