#!/bin/bash
# Slot trace of the pipelined kernel on the GPU box: builds libdfx with -DDFX_TRACE, prints tools/trace_slots.py's report
python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False, extra_flags=["-DDFX_TRACE"])
PY
python tools/trace_slots.py 2>&1 | grep -v "amdgpu.ids\|warning"
