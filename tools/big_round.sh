#!/bin/bash
# development: parity tests of the big scenes, then the exact-build bench lines of c5s and c4.  usage: tools/big_round.sh <tag> [pytest -k expr]
tag=$1; k=${2:-"c4_sparse or c5_sparse or big_scene or grid_search or random_scenes or planned_frames"}
timeout -k 10 600 python -u -m pytest tests -m gpu -q -x --timeout 300 --timeout-method thread -k "$k" > gpurun_out/${tag}_tests.log 2>&1 || { tail -30 gpurun_out/${tag}_tests.log; exit 1; }
tail -2 gpurun_out/${tag}_tests.log
for c in c5s c4; do
  timeout -k 10 200 python bench.py --config $c --numerics exact --no-cpu-baseline > gpurun_out/${tag}_bench_$c.json 2> gpurun_out/${tag}_bench_$c.err || { tail -5 gpurun_out/${tag}_bench_$c.err; exit 1; }
done
python - <<PY
import json
for c in ("c5s","c4"):
    d=json.load(open("gpurun_out/${tag}_bench_%s.json"%c)); print(c, "ms/step %.2f"%d["ms_per_step"], "alone %.2f"%d["roofline"]["one_frame_at_a_time"]["ms_per_step_one_frame_at_a_time"], "k0 %.2f"%d["roofline"]["one_frame_at_a_time"]["k_primary_ms"])
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in c5s c4; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${tag}_trace_$c -- python3 bench.py --config $c --numerics exact --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot --no-generic --shared-stream --inflight 1 --tail-stream off > gpurun_out/${tag}_trace_$c.log 2>&1 || { tail -5 gpurun_out/${tag}_trace_$c.log; exit 1; }
  echo "== $c"; python3 tools/big_trace.py gpurun_out/${tag}_trace_$c
done
if [ -f raytracer_amd/librt_hip_b.so ]; then
  for c in c5s c4; do
    RT_HIP_LIB=raytracer_amd/librt_hip_b.so timeout -k 10 200 python bench.py --config $c --numerics exact --no-cpu-baseline --no-generic > gpurun_out/${tag}_benchB_$c.json 2> gpurun_out/${tag}_benchB_$c.err || { tail -5 gpurun_out/${tag}_benchB_$c.err; exit 1; }
    python3 -c "import json;d=json.load(open('gpurun_out/${tag}_benchB_$c.json'));print('B build $c ms/step %.2f'%d['ms_per_step'])"
  done
fi
