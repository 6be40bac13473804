#!/bin/bash
# Round 2: the committed evidence, in one GPU call.  usage (GPU box, repo root): bash tools/final_profile_r02.sh <tag>
#  1. the DEFAULT bench command under rocprofv3 --kernel-trace --stats (+ the bench line of that same run)
#  2. PMC passes over short bench runs of BOTH builds (fast, exact), one rocprofv3 run per pass
#  3. tools/pmc_json.py turns the counter CSVs into profiles-ready JSON (gpurun_out/pmc_<tag>.json)
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }
grep "^{" gpurun_out/prof_$tag.log > gpurun_out/bench_under_prof_$tag.json
python3 tools/frame_timeline.py gpurun_out/prof_$tag > gpurun_out/timeline_$tag.txt
for num in fast exact; do
  n=0
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES" \
              "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH" \
              "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_LDS_CMD_FIFO_FULL" \
              "FETCH_SIZE" "WRITE_SIZE"; do
    n=$((n+1))
    timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_${tag}_${num}_$n -- python3 bench.py --steps 10 --warmup 1 --no-cpu-baseline --no-one-shot --no-generic --tail-stream off --numerics $num > gpurun_out/pmc_${tag}_${num}_$n.log 2>&1 || { tail -5 gpurun_out/pmc_${tag}_${num}_$n.log; exit 1; }
  done
done
python3 tools/pmc_json.py $tag > gpurun_out/pmc_$tag.json && head -c 1500 gpurun_out/pmc_$tag.json
