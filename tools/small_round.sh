#!/bin/bash
# development: parity suite, then the bench lines of config 2 (default) and of the 64-sphere scene (both builds).  usage: tools/small_round.sh <tag>
tag=$1
timeout -k 10 600 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread > gpurun_out/${tag}_tests.log 2>&1 || { tail -30 gpurun_out/${tag}_tests.log; exit 1; }
tail -1 gpurun_out/${tag}_tests.log
timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-one-shot > gpurun_out/${tag}_c2.json 2> gpurun_out/${tag}_c2.err || { tail -5 gpurun_out/${tag}_c2.err; exit 1; }
timeout -k 10 200 python bench.py --steps 60 --warmup 10 --numerics exact --no-cpu-baseline --no-one-shot > gpurun_out/${tag}_c2x.json 2> gpurun_out/${tag}_c2x.err || { tail -5 gpurun_out/${tag}_c2x.err; exit 1; }
for n in fast exact; do timeout -k 10 200 python bench.py --config syn64 --numerics $n --no-cpu-baseline --no-one-shot > gpurun_out/${tag}_syn64_$n.json 2> gpurun_out/${tag}_syn64_$n.err || { tail -5 gpurun_out/${tag}_syn64_$n.err; exit 1; }; done
python3 - <<PY
import json
for n in ("c2","c2x","syn64_fast","syn64_exact"):
    d=json.load(open("gpurun_out/${tag}_%s.json"%n)); r=d["roofline"]; o=r.get("one_frame_at_a_time",{}); g=d.get("generic_kernels",{})
    print("%-12s ms/step %.4f  k0 %.4f ms frac %.3f | alone k0 %.4f frac %.3f frame %.4f | generic k0 %.4f frac %.3f" % (n, d["ms_per_step"], r["avg_launch_ms"], r["frac"], o.get("k_primary_ms",0), o.get("frac",0), o.get("ms_per_step_one_frame_at_a_time",0), g.get("k_primary_ms",0), g.get("frac",0)))
PY
