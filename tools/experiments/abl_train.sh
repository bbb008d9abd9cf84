#!/bin/bash
# Timing ablations of the training feed-forward kernels on one box (per-kernel time under rocprofv3 for each flag set).
# The DFX_ABL_FF_* / DFX_FF_STAGGER variants are not part of the shipped kernels: git apply tools/patches/r03_train_ff_ablations.patch first.
# Results of round 3: profiles/r03_train_ff_ablations.txt
SETS=("" "-DDFX_ABL_FF_NODMA" "-DDFX_ABL_FF_NOBAR" "-DDFX_ABL_FF_NOGELU" "-DDFX_ABL_FF_NOEPI")
[ $# -gt 0 ] && SETS=("$@")
for F in "${SETS[@]}"; do
python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False, extra_flags="$F".split())
PY
echo "== flags [$F]"
tools/prof_train_kernels.sh gpurun_out/r03/train_kernel_stats_abl.csv 2>&1 | head -4
done
python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False)
PY
