show() { python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('%-46s %.4f ms/frame  k_primary %.4f  frac %.3f  alone %.4f' % (sys.argv[1], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline'].get('one_frame_at_a_time',{}).get('ms_per_step_one_frame_at_a_time',0)))" "$1"; }
B="timeout -k 10 120 python3 bench.py --no-cpu-baseline"
$B 2>/dev/null | show "default" || exit 1
RT_HIP_TAIL_LEVEL=2 $B --inflight 4 2>/dev/null | show "tail level 2, 4 in flight" || exit 1
RT_HIP_TAIL_LEVEL=2 $B --inflight 5 2>/dev/null | show "tail level 2, 5 in flight" || exit 1
RT_HIP_SUBTREE_MAX=16384 $B 2>/dev/null | show "subtree from level 2" || exit 1
RT_HIP_SUBTREE_MAX=16384 RT_HIP_TAIL_LEVEL=2 $B 2>/dev/null | show "subtree from level 2 on the tail" || exit 1
RT_HIP_SUBTREE_MAX=16384 RT_HIP_TAIL_LEVEL=2 $B --inflight 4 2>/dev/null | show "subtree from level 2 on the tail, 4 in flight" || exit 1
RT_HIP_SUBTREE_MAX=2048 $B 2>/dev/null | show "subtree max 2048" || exit 1
