#!/bin/bash
# development: grid density / light-grid budget / refill threshold sweeps on the big scenes (pipelined ms per frame, exact build)
for c in c4 c5s; do
  for d in 0.25 0.5 1 2; do
    RT_HIP_GRID_DENSITY=$d timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-generic --no-one-shot > gpurun_out/sw.json 2> gpurun_out/sw.err || { tail -3 gpurun_out/sw.err; exit 1; }
    python3 -c "import json;d=json.load(open('gpurun_out/sw.json'));print('$c density $d: %.2f ms'%d['ms_per_step'])"
  done
  for b in 4 8 32; do
    RT_HIP_LGRID_BUDGET=$b timeout -k 10 200 python bench.py --config $c --no-cpu-baseline --no-generic --no-one-shot > gpurun_out/sw.json 2> gpurun_out/sw.err || { tail -3 gpurun_out/sw.err; exit 1; }
    python3 -c "import json;d=json.load(open('gpurun_out/sw.json'));print('$c lgrid budget $b: %.2f ms'%d['ms_per_step'])"
  done
done
