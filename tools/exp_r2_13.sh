cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False, extra_flags=["-DDFX_TRACE"])
PY
python tools/trace_coop.py 2>&1 | grep -v "amdgpu.ids\|warning" > gpurun_out/r2/exp13.log
