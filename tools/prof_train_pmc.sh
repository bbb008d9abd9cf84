#!/bin/bash
# PMC counters of the training iteration's product kernels (separate --pmc passes, no tracing): tools/prof_train_pmc.sh <outdir>
set -u
OUT=$1
mkdir -p $OUT
export TMPDIR=/tmp
run() {
  local name=$1; shift
  rocprofv3 --pmc "$@" -d $OUT/$name --output-format csv -- python tools/bench_train.py 128 2048 bf16 > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" ${PMC_PAT:-k_gemm_bf16} > $OUT/$name.summary.txt 2>&1
  echo "== pass $name"; cat $OUT/$name.summary.txt
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA
