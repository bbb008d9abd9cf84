#!/bin/bash
# A/B of the exact-fp32 chain with the libm erf (shipped) against an Abramowitz-Stegun erf (git apply tools/patches/r03_f32_fast_erf.patch first):
# kernel time at T = 50 and the fp32 parity printouts.  Round 3: 213.4 vs 214.5 ms — no gain, so the exact path keeps libm's erf.
for F in "" "-DDFX_F32_FAST_ERF"; do
python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False, extra_flags="$F".split())
PY
echo "== flags [$F]"
python bench.py --precision f32 --timesteps 50 --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-train-line 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32 kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"
python -m pytest tests/test_gpu_denoiser.py tests/test_gpu_headline.py -m gpu -q -s -k "f32 or golden or oracle or headline_T1000 and contractive" 2>&1 | grep -E "passed|failed|FAILED|f32 vs oracle|max-abs" | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
done
python - <<PY
from difffacto_amd import build
build.build(force=True, verbose=False)
PY
