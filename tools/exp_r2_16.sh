cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
{
timeout 1200 python -m pytest tests/test_gpu_denoiser.py tests/test_gpu_modules.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3
for B in 1 2 4 8; do
 echo -n "B=$B: "
 python bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-train-line --no-parity 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f shapes/s  %.2f ms per batch  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"
done
python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False, extra_flags=["-DDFX_TRACE"])
PY
python tools/trace_coop.py 2>&1 | grep -v "amdgpu.ids\|warning"
} > gpurun_out/r2/exp16.log 2>&1
