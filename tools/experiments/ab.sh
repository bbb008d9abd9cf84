#!/bin/bash
# A/B timing of two builds of libdfx on the same box: tools/experiments/ab.sh "<extra flags A>" "<extra flags B>" [bench args]
A="$1"; B="$2"; shift 2
ARGS=${*:---timesteps 50 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-train-line}
for v in A B A B; do
  if [ $v = A ]; then F="$A"; else F="$B"; fi
  python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False, extra_flags="$F".split())
PY
  echo -n "$v [$F]: "
  python bench.py $ARGS 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms %.3f  frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
