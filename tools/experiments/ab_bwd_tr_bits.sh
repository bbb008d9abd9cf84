cd $GRAFT_REPO_ROOT
for arm in "tr0:-DDFX_FF_BWD_TR=0" "tr1:"; do
  name=${arm%%:*}; flags=${arm#*:}
  touch difffacto_amd/csrc/train_kernels.hip
  python -c "import sys; from difffacto_amd import build; build.build(verbose=False, extra_flags=sys.argv[1].split())" "$flags" || exit 1
  python tools/experiments/dump_train_step.py /tmp/${name}_a.npz 16 2048 0.2
  python tools/experiments/dump_train_step.py /tmp/${name}_b.npz 3 160
  python tools/experiments/dump_train_step.py /tmp/${name}_c.npz 64 2048 0.2
done
for s in a b c; do python tools/experiments/dump_train_step.py --cmp /tmp/tr0_$s.npz /tmp/tr1_$s.npz; done
