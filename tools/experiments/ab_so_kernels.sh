#!/bin/bash
# Same-box A/B of two prebuilt libraries, per-kernel times of the training iteration: tools/experiments/ab_so_kernels.sh [bench_train args]
D=difffacto_amd
for v in base new base new; do
  cp $D/_ab/libdfx_$v.so $D/libdfx.so
  echo "== $v: $(python tools/bench_train.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-130)"
done
for v in base new; do
  cp $D/_ab/libdfx_$v.so $D/libdfx.so
  echo "== $v (kernel trace)"
  tools/prof_train_kernels.sh gpurun_out/ab_$v.csv | grep -E "k_ff|k_attn_bwd_param|k_ln3|in all" | cut -c1-120
done
cp $D/_ab/libdfx_new.so $D/libdfx.so
