#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_round.sh <tag> [bench args...]  - four PMC passes over the bench, one run each
tag=$1; shift
for pass in "a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES" \
            "b SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" \
            "c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH"; do
  set -- $pass_args
  p=($pass); name=${p[0]}; ctrs="${p[@]:1}"
  bash tools/pmc_bench.sh ${tag}_$name "$BENCH_ARGS" $ctrs || exit 1
done
