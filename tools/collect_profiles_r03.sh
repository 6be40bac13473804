#!/bin/bash
# copies what tools/final_profile_r03.sh <tag> left under gpurun_out/ (scratch) into profiles/ (tracked).  usage: tools/collect_profiles_r03.sh <tag>
tag=${1:-r03}
cd "$(dirname "$0")/.."
cp gpurun_out/pmc_$tag.json profiles/r03_pmc.json
cp "$(ls -S gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1)" profiles/r03_kernel_stats_default_bench.csv
python3 tools/timed_region_stats.py gpurun_out/prof_$tag 200 30 > profiles/r03_k_primary_timed_region.txt
cp gpurun_out/bench_line_$tag.json profiles/r03_bench_line.json
cp gpurun_out/bench_under_prof_$tag.json profiles/r03_bench_line_under_rocprof.json
python3 tools/frame_timeline.py gpurun_out/prof_$tag > profiles/r03_frame_timeline.txt
cp gpurun_out/hbm_by_kernel_$tag.txt profiles/r03_hbm_by_kernel.txt
for c in c4 c5s; do
  cp gpurun_out/big_${tag}_$c.txt profiles/r03_e_big_${c}_final.txt
  cp "$(ls gpurun_out/prof_${tag}_$c/*/*kernel_stats.csv | head -1)" profiles/r03_e_kernel_stats_${c}_final.csv
done
cp gpurun_out/bench_configs_$tag.jsonl profiles/r03_bench_configs.jsonl
ls -la profiles | grep r03
