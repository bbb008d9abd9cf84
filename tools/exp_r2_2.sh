cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
{
bash tools/ab.sh "" "-DDFX_GELU_POLY"
bash tools/run_trace.sh -DDFX_GELU_POLY
python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False, extra_flags=["-DDFX_GELU_POLY"])
PY
python -m pytest tests/test_gpu_denoiser.py -x -q -m gpu 2>&1 | tail -5
python tools/report_parity.py 2>&1 | tail -30
} 2>&1 | grep -v "warning\|^ *[0-9]* |\|^ *|\|mfma_linear\|In file included\|generated" > gpurun_out/r2/exp2.log
