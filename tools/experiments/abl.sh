#!/bin/bash
# Timing ablations of the chain kernel on one box: tools/experiments/abl.sh "<common flags>" "<flags 1>" "<flags 2>" ...
# The DFX_ABL_* / DFX_SWP variants are not part of the shipped kernel: apply tools/patches/r02_pipe_kernel_ablations_and_swp.patch first
# (git apply tools/patches/...; the patch is against the round-2 kernel before the workgroup-size templating and may need a 3-way merge).
C="$1"; shift
for F in "" "$@"; do
  python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False, extra_flags="$C $F".split())
PY
  echo -n "[$C $F]: "
  python bench.py --timesteps 50 --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-train-line 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms %.3f' % d['roofline']['kernel_ms'])"
done
