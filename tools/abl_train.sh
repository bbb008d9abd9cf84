#!/bin/bash
for F in "-DDFX_ABL_FF_NOEPI" "-DDFX_ABL_FF_NOEPI -DDFX_ABL_FF_NOBAR -DDFX_ABL_FF_NOGELU -DDFX_ABL_FF_NODMA"; do
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
