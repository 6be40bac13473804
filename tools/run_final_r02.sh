bash tools/final_profile_r02.sh r02f > gpurun_out/final_r02f.log 2>&1 || { tail -5 gpurun_out/final_r02f.log; exit 1; }
cd /tmp && export TMPDIR=/tmp && timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_k0f -- python3 $GRAFT_REPO_ROOT/tools/k0_time.py fast > /dev/null 2>&1 || exit 1
cd $GRAFT_REPO_ROOT && python3 tools/frame_timeline.py gpurun_out/prof_k0f > gpurun_out/timeline_k0f.txt
timeout -k 10 300 python3 bench.py > gpurun_out/bench_final.log 2>&1 || exit 1
grep "^{" gpurun_out/bench_final.log > gpurun_out/bench_final.json
for c in c1 c3 c4; do timeout -k 10 300 python3 bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/bench_cfgs.jsonl || exit 1; done
timeout -k 10 300 python3 bench.py --config c5s --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/bench_cfgs.jsonl || exit 1
timeout -k 10 300 python3 bench.py --config c5s --steps 6 --warmup 1 --no-cpu-baseline --numerics exact 2>/dev/null | tail -1 >> gpurun_out/bench_cfgs.jsonl || exit 1
timeout -k 10 300 python3 bench.py --numerics exact --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/bench_cfgs.jsonl || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/bench_cfgs.jsonl || exit 1
timeout -k 10 300 python3 bench.py --launch-plan off --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/bench_cfgs.jsonl || exit 1
python3 -c "
import json
for l in open('gpurun_out/bench_cfgs.jsonl'):
    d=json.loads(l); print(d['config']['workload'][:30], d['config']['numerics'], d['config'].get('k_primary_launch_plan'), d['steps'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])
d=json.load(open('gpurun_out/bench_final.json')); print('final', d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['one_frame_at_a_time'], d['cpu_baseline'])
"
