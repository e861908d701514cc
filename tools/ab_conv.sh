#!/bin/bash
# In-box A/B of two builds of libhsidm.so on single conv launches (tools/conv_bench.py, batch 240, sustained):
#   PREC=fp16x1 bash tools/ab_conv.sh OTHER.so SHAPE [SHAPE ...]      (alternates base / other, two rounds)
other=$1; shift
for r in 1 2; do
  for lib in "" "$other"; do
    if [ -z "$lib" ]; then tag=new; unset HSIDM_LIB; else tag=$(basename $lib); export HSIDM_LIB=$PWD/$lib; fi
    for s in "$@"; do
      python tools/conv_bench.py --batch ${BATCH:-240} --precision ${PREC:-fp16x1} --only $s --sustain 1.5 2>/dev/null | grep sustained | sed "s/^/$tag /"
    done
  done
done
