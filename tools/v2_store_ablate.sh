#!/bin/bash
# What do the output stores of the persistent 3x3 kernel cost?  Diagnostic build (-DHSIDM_V2_ABLATE), HSIDM_V2_ABL=1 drops them (results
# are wrong; timing only).  Run on the GPU box from the repo root:  bash tools/v2_store_ablate.sh
HSIDM_EXTRA_FLAGS="-DHSIDM_V2_ABLATE" HSIDM_OUT=$PWD/gpurun_out/libabl.so HSIDM_OBJ=/tmp/obj_abl bash hsi-dmgasr_amd/csrc/build.sh > /dev/null 2>&1
export HSIDM_LIB=$PWD/gpurun_out/libabl.so
for m in 0 1 0 1; do
  echo "V2_ABL=$m"
  for s in l64_128_128 l32_256_256 l32_768_256 l16_512_512 l16_1024_512; do
    HSIDM_V2_ABL=$m python tools/conv_bench.py --batch 240 --reps 7 --only $s 2>/dev/null | grep -v amdgpu
  done
done
rm -f gpurun_out/libabl.so
