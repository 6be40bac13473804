#!/bin/bash
# Round 3: what one rank of an N-GPU run does, rehearsed on the one GPU of the box (same code path as under
# torch.distributed.run, the all-gather left out): config 3 at N = 2, 4, 8, config 2 at N = 8, full-size config 5 at N = 8.
show() { python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); c = d['config']
print('%-28s rank share %.4f ms/frame  k_primary %.4f ms  numerics %s  %s in flight %s  device MiB %s' % (sys.argv[1], d['ms_per_step'], d['roofline']['avg_launch_ms'], c['numerics'], c.get('streams'), c.get('frames_in_flight'), c.get('device_mib')))" "$1"; }
B="timeout -k 10 280 python3 bench.py --no-cpu-baseline --no-generic --no-one-shot"
for n in 2 4 8; do
  RT_BENCH_REHEARSE_WORLD=$n RT_BENCH_REHEARSE_RANK=1 $B --gpus $n --config c3 2>/dev/null | show "c3 N=$n rank 1" || exit 1
done
RT_BENCH_REHEARSE_WORLD=8 RT_BENCH_REHEARSE_RANK=3 $B --gpus 8 --config c2 2>/dev/null | show "c2 N=8 rank 3" || exit 1
RT_BENCH_REHEARSE_WORLD=8 RT_BENCH_REHEARSE_RANK=3 $B --gpus 8 --config c5 --steps 2 --warmup 1 2>gpurun_out/n8_c5.err | show "c5 (full size) N=8 rank 3" || { tail -5 gpurun_out/n8_c5.err; exit 1; }
