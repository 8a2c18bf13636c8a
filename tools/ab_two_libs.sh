#!/bin/bash
# A/B of two BUILDS of the library on one box: alternating processes, the engine-level step and the per-op table of each.
#   tools/ab_two_libs.sh <libA.so> <libB.so> <outdir> [rounds]
A=$1; B=$2; O=$3; R=${4:-3}
mkdir -p $O
for i in $(seq 1 $R); do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    AEW_LIB_PATH=$PWD/$lib python tools/ab_step.py base --rounds 3 --steps 20 --per-op base --out $O/$v$i > $O/$v$i.log 2>&1
    echo "$v$i $(grep -h '^base' $O/$v$i.log)  dz: $(grep -h ' dz\.#' $O/$v$i/per_op_base.txt)  dx: $(grep -h ' dx\.#' $O/$v$i/per_op_base.txt)"
  done
done
