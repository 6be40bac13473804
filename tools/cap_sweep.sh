#!/bin/bash
# development: what a lower limit on a self-sizing context's queues / fewer frames in flight cost the big frames
show() { python3 -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); c = d['config']
print('%-36s %.3f ms/frame  device MiB per context %s, all %s, batches %s retries %s' % (sys.argv[1], d['ms_per_step'], c.get('device_mib_per_context'), c.get('device_mib'), c.get('batches_per_frame'), c.get('overflow_retries')))" "$1"; }
B="timeout -k 10 280 python3 bench.py --no-cpu-baseline --no-generic --no-one-shot"
for cap in 0 33554432; do
  for inf in 2 3; do
    RT_HIP_QUEUE_CAP=$cap $B --config c4 --inflight $inf 2>/dev/null | show "c4 limit $cap inflight $inf" || exit 1
  done
done
RT_HIP_QUEUE_CAP=33554432 $B --config c5s --inflight 3 2>/dev/null | show "c5s limit 32Mi inflight 3" || exit 1
$B --config c5s --inflight 3 2>/dev/null | show "c5s inflight 3" || exit 1
