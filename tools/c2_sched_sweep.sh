#!/bin/bash
# development: config 2 under other stream arrangements (ms per frame, event-timed k_primary)
for lv in 1 2 3 4; do
for a in "--render-priority normal" "--render-priority normal --inflight 4" "--render-priority high"; do
  RT_HIP_TAIL_LEVEL=$lv timeout -k 10 200 python bench.py $a --no-cpu-baseline --no-one-shot --no-generic > gpurun_out/sw.json 2> gpurun_out/sw.err || { tail -3 gpurun_out/sw.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/sw.json'));print('tail level $lv %-40s %.4f ms/frame  k_primary %.4f ms  p50 %.4f p95 %.4f'%('$a' or 'default', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['step_ms']['p50'], d['step_ms']['p95']))"
done; done
