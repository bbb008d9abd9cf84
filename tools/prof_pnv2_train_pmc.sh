#!/bin/bash
# PMC counters of the PointNetV2 training path's fp32 product kernels (separate --pmc passes): tools/prof_pnv2_train_pmc.sh <outdir> [kernel name pattern]
set -u
OUT=$1; PAT=${2:-k_lin_wide_lds}
mkdir -p $OUT
export TMPDIR=/tmp
run() {
  local name=$1; shift
  rocprofv3 --pmc "$@" -d $OUT/$name --output-format csv -- python tools/experiments/time_pointnet_v2_train.py 128 2048 > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" $PAT > $OUT/$name.summary.txt 2>&1
  echo "== pass $name"; cat $OUT/$name.summary.txt
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
