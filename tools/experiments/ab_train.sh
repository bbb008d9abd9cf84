#!/bin/bash
# A/B of two builds on the training iteration: tools/experiments/ab_train.sh "<flags A>" "<flags B>"
for F in "$1" "$2" "$1" "$2"; do
  python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False, extra_flags="$F".split())
PY
  echo -n "[$F]: "; python tools/bench_train.py 128 2048 bf16 2>&1 | tail -1 | cut -c1-120
done
