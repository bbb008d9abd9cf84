#!/bin/bash
# Same-box A/B of chain-kernel variants of ONE build: tools/experiments/ab_variants.sh "<pipe-waves A>" "<pipe-waves B>" [bench args]
A="$1"; B="$2"; shift 2
ARGS=${*:---timesteps 50 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-train-line}
for v in $A $B $A $B; do
  echo -n "pipe-waves $v: "
  python bench.py $ARGS --pipe-waves $v 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms %.3f  frac %.4f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))"
done
