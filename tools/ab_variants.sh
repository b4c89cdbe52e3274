#!/bin/bash
# Build kernel variants (compile-time knobs) side by side and bench each in its own process on one GPU.
# usage: tools/ab_variants.sh "name1:-DFLAG=1 -DX=2" "name2:..."   (run on the GPU box; build happens wherever called)
set -e
cd "$(dirname "$0")/.."
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  make -s -C deepchopper_amd/csrc OBJDIR=../lib/obj_$name OUT=../lib/libdc_$name.so EXTRA="$flags" -j8 >/dev/null
done
