#!/bin/bash
# Are two builds of the training kernels bit-identical?  (GPU box)  usage: ab_bits.sh "-DFLAGS_OF_ARM_0" ["-DFLAGS_OF_ARM_1"]   (default arm 1 = the shipped build)
# eps + 79 gradients of one iteration at three shapes (16 x 2048 with Dropout 0.2, 3 x 160, 64 x 2048 with Dropout 0.2) from each build, compared bit by bit.
# Used for: k_ff<true>'s transpose-read ring (-DDFX_FF_BWD_TR=0 against the default), k_ff_wgrad's transpose-read consumers (-DDFX_WG_TR=0).
cd ${GRAFT_REPO_ROOT:-/root/repo}
A0=${1:--DDFX_FF_BWD_TR=0}; A1=${2:-}
i=0
for flags in "$A0" "$A1"; do
  touch difffacto_amd/csrc/train_kernels.hip
  python -c "import sys; from difffacto_amd import build; build.build(verbose=False, extra_flags=sys.argv[1].split())" "$flags" || exit 1
  python tools/experiments/dump_train_step.py /tmp/arm${i}_a.npz 16 2048 0.2 | grep -v amdgpu
  python tools/experiments/dump_train_step.py /tmp/arm${i}_b.npz 3 160 | grep -v amdgpu
  python tools/experiments/dump_train_step.py /tmp/arm${i}_c.npz 64 2048 0.2 | grep -v amdgpu
  i=$((i+1))
done
for s in a b c; do python tools/experiments/dump_train_step.py --cmp /tmp/arm0_$s.npz /tmp/arm1_$s.npz; done
touch difffacto_amd/csrc/train_kernels.hip; python -c "from difffacto_amd import build; build.build(verbose=False)"
