#!/bin/bash
# Round 3: the committed evidence, in one GPU call.  usage (GPU box, repo root): bash tools/final_profile_r03.sh [tag]
#  1. the DEFAULT bench command under rocprofv3 --kernel-trace --stats (+ the bench line of that same run), and unprofiled
#  2. PMC passes over short bench runs of BOTH builds (fast, exact), one rocprofv3 run per pass -> gpurun_out/pmc_<tag>.json
#  3. HBM bytes per kernel of a frame (tools/hbm_by_kernel.sh)
#  4. the big scenes: kernel trace + PMC table per kernel (tools/profile_big_r03.sh)
#  5. one bench line per other configuration and build
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }
grep "^{" gpurun_out/prof_$tag.log > gpurun_out/bench_under_prof_$tag.json
python3 tools/frame_timeline.py gpurun_out/prof_$tag > gpurun_out/timeline_$tag.txt
timeout -k 10 300 python3 bench.py > gpurun_out/bench_line_$tag.json 2> gpurun_out/bench_line_$tag.err || { tail -5 gpurun_out/bench_line_$tag.err; exit 1; }
for num in fast exact; do
  n=0
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES" \
              "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH" \
              "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F64" \
              "SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
              "FETCH_SIZE" "WRITE_SIZE"; do
    n=$((n+1))
    timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_${tag}_${num}_$n -- python3 bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-one-shot --no-generic --tail-stream off --numerics $num > gpurun_out/pmc_${tag}_${num}_$n.log 2>&1 || { tail -5 gpurun_out/pmc_${tag}_${num}_$n.log; exit 1; }
  done
done
python3 tools/pmc_json_r03.py $tag > gpurun_out/pmc_$tag.json || exit 1
bash tools/hbm_by_kernel.sh > gpurun_out/hbm_by_kernel_$tag.txt || exit 1
bash tools/profile_big_r03.sh $tag c4 c5s > gpurun_out/profile_big_$tag.log 2>&1 || { tail -5 gpurun_out/profile_big_$tag.log; exit 1; }
: > gpurun_out/bench_configs_$tag.jsonl
for a in "--config c1" "--config c3" "--config c4" "--config c5s" "--config syn64 --numerics fast" "--config syn64" "--numerics exact" "--launch-plan off"; do
  timeout -k 10 300 python3 bench.py $a --no-cpu-baseline --no-one-shot --no-small-scene >> gpurun_out/bench_configs_$tag.jsonl 2> gpurun_out/bench_configs_$tag.err || { echo "bench.py $a"; tail -5 gpurun_out/bench_configs_$tag.err; exit 1; }
done
echo done
