#!/bin/bash
python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False, extra_flags=["-DDFX_TRACE"])
PY
python tools/experiments/trace_f32.py 2>&1 | grep -v "amdgpu.ids\|warning"
python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False)
PY
