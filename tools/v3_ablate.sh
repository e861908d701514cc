#!/bin/bash
# What is each phase of conv_v3's work item worth?  Diagnostic build (-DHSIDM_V2_ABLATE; tools/_libabl.so, built here if absent),
# HSIDM_V2_ABL masks (results are wrong; timing only): 2 commit without GroupNorm + SiLU (raw copy), 16 no commit, 4 no halo loads,
# 32 no per-item address computation (the first item's addresses), 1 no output stores, 64 no epilogue, 8 no weight loads, 128 no MFMAs.
# Run on the GPU box from the repo root:  bash tools/v3_ablate.sh [PREC] [SHAPE ...]
prec=${1:-fp16x1}; shift
shapes=${@:-l128_64_64 l128_192_64}
[ -f tools/_libabl.so ] || HSIDM_EXTRA_FLAGS="-DHSIDM_V2_ABLATE" HSIDM_OUT=$PWD/tools/_libabl.so HSIDM_OBJ=/tmp/obj_abl bash hsi-dmgasr_amd/csrc/build.sh > /dev/null 2>&1
export HSIDM_LIB=$PWD/tools/_libabl.so HSIDM_EXPERIMENTAL=1
for m in 0 2 16 4 32 36 1 64 8 128 0 2 50 118; do
  for s in $shapes; do
    HSIDM_V2_ABL=$m python tools/conv_bench.py --batch 240 --reps 5 --precision $prec --only $s --sustain 1.0 2>/dev/null | grep sustained | sed "s/^/abl=$m /"
  done
done
