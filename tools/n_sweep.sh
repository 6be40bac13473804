#!/bin/bash
# development: a rank of N on config 3 (rehearsal): frames on their own streams, how many in flight
show() { python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('%-40s %.4f ms/frame  k_primary %.4f' % (sys.argv[1], d['ms_per_step'], d['roofline']['avg_launch_ms']))" "$1"; }
for n in 8 4 2; do
  export RT_BENCH_REHEARSE_WORLD=$n
  B="timeout -k 10 120 python3 bench.py --config c3 --no-cpu-baseline --no-generic"
  $B 2>/dev/null | show "N=$n default" || exit 1
  for f in 4 6 8; do $B --overlap --inflight $f 2>/dev/null | show "N=$n overlap $f" || exit 1; done
done
