#!/bin/bash
# development: a rank of eight on config 3 (rehearsal), pipelining options
show() { python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('%-40s %.4f ms/frame  k_primary %.4f' % (sys.argv[1], d['ms_per_step'], d['roofline']['avg_launch_ms']))" "$1"; }
export RT_BENCH_REHEARSE_WORLD=8 RT_BENCH_REHEARSE_RANK=1
B="timeout -k 10 120 python3 bench.py --gpus 8 --config c3 --no-cpu-baseline --no-generic --no-one-shot"
$B 2>/dev/null | show "default (stream per frame, 6)" || exit 1
$B --inflight 4 2>/dev/null | show "stream per frame, 4" || exit 1
$B --inflight 8 2>/dev/null | show "stream per frame, 8" || exit 1
$B --inflight 12 2>/dev/null | show "stream per frame, 12" || exit 1
$B --render-priority normal 2>/dev/null | show "equal priorities, 6" || exit 1
$B --shared-stream 2>/dev/null | show "shared + tail stream, 3" || exit 1
$B --shared-stream --inflight 5 2>/dev/null | show "shared + tail stream, 5" || exit 1
$B --gather-stream same 2>/dev/null | show "gather on the render stream" || exit 1
RT_BENCH_HOST_TIMES=1 $B 2>&1 | grep -i "host" | tail -2
unset RT_BENCH_REHEARSE_WORLD RT_BENCH_REHEARSE_RANK
timeout -k 10 120 python3 bench.py --config c3 --no-cpu-baseline --no-generic --no-one-shot 2>/dev/null | show "c3 on ONE GPU" || exit 1
