#!/bin/bash
# development A/B: bench the ablation builds under tools/abl (wrong pixels, timing only)
for lib in "" "$@"; do
  RT_HIP_LIB=${lib:+$PWD/$lib} timeout -k 10 200 python3 -u bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | \
   python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-baseline}', 'step_ms', round(d['ms_per_step'],4), 'k0_ms', round(d['roofline']['avg_launch_ms'],4), 'rays', d['config']['rays_per_step'])"
done
