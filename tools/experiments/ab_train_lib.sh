#!/bin/bash
# Same-box A/B of two BUILT libraries on the training iteration (GPU box): tools/experiments/ab_train_lib.sh <base libdfx.so> [rounds]
# Per-kernel times of tools/bench_train.py under rocprofv3 --kernel-trace --stats, base / new alternating (the new one = difffacto_amd/libdfx.so).
BASE=$1; ROUNDS=${2:-2}
L=difffacto_amd/libdfx.so
cp $L /tmp/libdfx_new.so
for r in $(seq $ROUNDS); do
  for v in base new; do
    if [ $v = base ]; then cp $BASE $L; else cp /tmp/libdfx_new.so $L; fi
    echo "== $v (round $r): $(python tools/bench_train.py | tail -1 | cut -c1-140)"
    tools/prof_train_kernels.sh /tmp/ab_$v.csv 2>&1 | head -${AB_LINES:-6}
  done
done
cp /tmp/libdfx_new.so $L
