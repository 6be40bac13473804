#!/bin/bash
# one full GPU round as the driver runs it at round end: the GPU suite, smoke(), the default bench
timeout -k 10 900 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/full_tests.log 2>&1
rc=$?; tail -3 gpurun_out/full_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
timeout -k 10 300 python3 bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err || { tail -5 gpurun_out/full_bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/full_bench.json')); r=d['roofline']
print('bench: %.4f ms/frame, %.1f Grays/s, k_primary %.4f ms frac %.3f (alone %.4f, %.3f); generic %.3f run-time %.3f; syn64 %.3f; traffic %s fp_exec %s' % (d['ms_per_step'], d['value']/1e3, r['avg_launch_ms'], r['frac'], r['one_frame_at_a_time']['k_primary_ms'], r['one_frame_at_a_time']['frac'], d['generic_kernels']['frac'], d['run_time_kernels']['frac'], d['small_scene']['frac'], r.get('traffic'), r.get('fp_executed_frac')))"
