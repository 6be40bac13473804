#!/bin/bash
# Round 3: rocprofv3 evidence for the big-scene (sphere grid) kernels.  usage (GPU box, repo root):
#   bash tools/profile_big_r03.sh <tag> [configs...]          default configs: c4 c5s
#  1. bench.py --config <cfg> --numerics exact, frames one at a time, under --kernel-trace --stats
#  2. PMC passes over the same command (one rocprofv3 run per pass)
#  3. tools/pmc_big.py: per-kernel table (time, waves, VALU busy, active lanes, waits, cache hit rates)
tag=${1:-r03}; shift
cfgs=${@:-c4 c5s}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in $cfgs; do
  cmd="bench.py --config $cfg --numerics exact --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot --no-generic --shared-stream --inflight 1 --tail-stream off"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$cfg -- python3 $cmd > gpurun_out/prof_${tag}_$cfg.log 2>&1 || { tail -5 gpurun_out/prof_${tag}_$cfg.log; exit 1; }
  n=0
  for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
              "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_FMA_F32" \
              "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
              "FETCH_SIZE" "WRITE_SIZE"; do
    n=$((n+1))
    timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d gpurun_out/pmc_${tag}_${cfg}_$n -- python3 $cmd > gpurun_out/pmc_${tag}_${cfg}_$n.log 2>&1 || { tail -5 gpurun_out/pmc_${tag}_${cfg}_$n.log; exit 1; }
  done
  python3 tools/pmc_big.py $tag $cfg > gpurun_out/big_${tag}_$cfg.txt || exit 1
  cat gpurun_out/big_${tag}_$cfg.txt
done
