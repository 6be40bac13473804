#!/bin/bash
# development: config 2 with k_subtree taking over at other level sizes (RT_HIP_SUBTREE_MAX hits per shard) and tail cuts
for sm in 4096 8192 16384 65536; do
for lv in 2 3; do
  RT_HIP_SUBTREE_MAX=$sm RT_HIP_TAIL_LEVEL=$lv timeout -k 10 200 python bench.py --no-cpu-baseline --no-one-shot --no-generic > gpurun_out/sw.json 2> gpurun_out/sw.err || { tail -3 gpurun_out/sw.err; exit 1; }
  python3 -c "import json;d=json.load(open('gpurun_out/sw.json'));print('subtree max $sm tail level $lv: %.4f ms/frame  k_primary %.4f ms  p50 %.4f p95 %.4f'%(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['step_ms']['p50'], d['step_ms']['p95']))"
done; done
