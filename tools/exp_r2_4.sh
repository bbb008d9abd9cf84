cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
{
bash tools/ab.sh "" "-DDFX_SWP"
python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False, extra_flags=["-DDFX_SWP"])
PY
python -m pytest tests/test_gpu_denoiser.py tests/test_gpu_headline.py -x -q -m gpu -s 2>&1 | grep -v amdgpu.ids | tail -25
} 2>&1 | grep -v "warning\|^ *[0-9]* |\|^ *|\|mfma_linear\|In file included\|generated" > gpurun_out/r2/exp4.log
