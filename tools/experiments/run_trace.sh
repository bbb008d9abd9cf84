#!/bin/bash
# Slot trace of the pipelined kernel on the GPU box: builds libdfx with -DDFX_TRACE (+ extra flags given as arguments),
# prints tools/experiments/trace_slots.py's report.  tools/experiments/run_trace.sh [-DDFX_ABL_NO_GELU ...]
python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False, extra_flags=["-DDFX_TRACE"] + "$*".split())
PY
python tools/experiments/trace_slots.py 2>&1 | grep -v "amdgpu.ids\|warning"
