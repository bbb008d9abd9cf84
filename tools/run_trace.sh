python -c "
from difffacto_amd import build
build.build(force=True, verbose=False, extra_flags=['-DDFX_TRACE'])
"
python tools/trace_slots.py 2>&1 | tail -12
