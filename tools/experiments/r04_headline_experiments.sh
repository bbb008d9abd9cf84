#!/bin/bash
# same-box A/B of two builds of the headline kernel against the shipped one (GPU box).
#   -DDFX_EXP_LN_MFMA   LayerNorm statistics (sum, sum of squares) on the matrix pipe (ones x h, h^T h)
#   -DDFX_EXP_B1_FOLD   no b1 accumulator initialisers: channel 127's K slot carries the constant 1, b1' rides in W1 — measured with this script
#                       in round 4 (profiles/r04_headline_experiments.txt) and SHIPPED since (bias_slot_one in denoiser_kernel.hip): the flag is now a no-op
# -DDFX_EXP_LN_MFMA needs tools/patches/r04_ln_stats_on_mfma.patch applied (git apply; the #ifdef block is not in the shipped sources).
ARGS="--timesteps 50 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-train-line"
for round in 1 2; do
  for F in "" "-DDFX_EXP_LN_MFMA" "-DDFX_EXP_B1_FOLD"; do
    python -c "from difffacto_amd import build; build.build(force=True, verbose=False, extra_flags='$F'.split())"
    echo -n "[${F:-shipped}] round $round: "
    python bench.py $ARGS 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('kernel_ms %.3f  frac %.4f  %s' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['kernel_variant']))"
    if [ $round = 1 ]; then python tools/experiments/exp_parity.py 100 32 2>&1 | tail -1; fi
  done
done
python -c "from difffacto_amd import build; build.build(force=True, verbose=False)"
