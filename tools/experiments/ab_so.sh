#!/bin/bash
# Same-box A/B of two prebuilt libraries: tools/experiments/ab_so.sh <command...>   (difffacto_amd/_ab/libdfx_base.so vs libdfx_new.so, A B A B)
D=difffacto_amd
for v in base new base new; do
  cp $D/_ab/libdfx_$v.so $D/libdfx.so
  echo -n "$v: "
  "$@" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-160
done
cp $D/_ab/libdfx_new.so $D/libdfx.so
