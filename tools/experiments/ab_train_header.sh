#!/bin/bash
# Same-box A/B of two versions of csrc/train_ff_fused.h (GPU box): tools/experiments/ab_train_header.sh <base header> [rounds]
# Per-kernel times of tools/bench_train.py under rocprofv3 --kernel-trace --stats, base / new alternating.
BASE=$1; ROUNDS=${2:-2}
H=difffacto_amd/csrc/train_ff_fused.h
cp $H /tmp/new_header.h
for r in $(seq $ROUNDS); do
  for v in base new; do
    if [ $v = base ]; then cp $BASE $H; else cp /tmp/new_header.h $H; fi
    python -c "from difffacto_amd import build; build.build(force=True, verbose=False)"
    echo "== $v (round $r): $(python tools/bench_train.py | tail -1 | cut -c1-140)"
    tools/prof_train_kernels.sh /tmp/ab_$v.csv 2>&1 | head -5
  done
done
cp /tmp/new_header.h $H
python -c "from difffacto_amd import build; build.build(force=True, verbose=False)"
