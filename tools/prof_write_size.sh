#!/bin/bash
# Does the chain kernel's WRITE_SIZE scale with T?  ("2.0 MB/step of WRITE_SIZE from a kernel that stores nothing per step is unexplained".)
# Separate rocprofv3 --pmc passes (never combined with tracing) of the headline command at T = 20 and T = 40, B = 128 x 2048:
#   WRITE_SIZE, and the request-level split TCC_EA0_WRREQ_sum / TCC_EA0_WRREQ_64B_sum / TCC_EA0_ATOMIC_sum.
# usage: tools/prof_write_size.sh <outdir>
set -u
REPO=$PWD
OUT=$(realpath -m $1); mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for T in 20 40; do
  for C in "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_STALL_sum" "FETCH_SIZE"; do
    tag=T${T}_$(echo $C | tr ' ' '+')
    rocprofv3 --pmc $C -d $OUT/$tag --output-format csv -- python $REPO/bench.py --timesteps $T --steps 3 --warmup 1 --no-parity --no-cpu-baseline --no-train-line > $OUT/$tag.log 2>&1
    f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
    echo "== T=$T  $C"
    [ -n "$f" ] && python $REPO/tools/pmc_summary.py "$f" k_denoise_pipe
  done
done
