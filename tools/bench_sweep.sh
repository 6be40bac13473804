#!/bin/bash
# development: bench.py's pipelining options side by side (ms per frame, k_primary ms in the timed region)
show() { python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('%-46s %.4f ms/frame  k_primary %.4f  frac %.3f' % (sys.argv[1], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))" "$1"; }
B="timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-generic"
$B 2>/dev/null | show "default" || exit 1
$B --inflight 2 2>/dev/null | show "inflight 2" || exit 1
$B --inflight 4 2>/dev/null | show "inflight 4" || exit 1
$B --render-priority normal 2>/dev/null | show "equal priorities" || exit 1
RT_HIP_TAIL_LEVEL=2 $B 2>/dev/null | show "tail level 2" || exit 1
RT_HIP_TAIL_LEVEL=1 $B 2>/dev/null | show "tail level 1" || exit 1
$B --tail-stream off 2>/dev/null | show "no tail stream" || exit 1
$B --overlap 2>/dev/null | show "overlap 3" || exit 1
$B --overlap --inflight 2 2>/dev/null | show "overlap 2" || exit 1
