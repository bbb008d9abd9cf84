#!/bin/bash
# Collect rocprofv3 PMC counters for the chain kernel (separate passes; --pmc never combined with tracing).
# usage: tools/prof_pmc.sh <outdir> [bench args...]
set -u
REPO=$PWD
OUT=$(realpath -m $1); shift
mkdir -p $OUT
cd /tmp   # rocprofv3 wants a writable cwd / TMPDIR
export TMPDIR=/tmp
run() { # name, counters
  local name=$1; shift
  rocprofv3 --pmc "$@" -d $OUT/$name --output-format csv -- python $REPO/bench.py --no-cpu-baseline $BENCH_ARGS > $OUT/$name.log 2>&1
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $REPO/tools/pmc_summary.py "$f" k_denoise > $OUT/$name.summary.txt 2>&1
  cat $OUT/$name.summary.txt
}
BENCH_ARGS="$*"
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
