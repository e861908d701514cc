#!/bin/bash
# Is the chip power-limited under these kernels?  The SAME launch, back to back for ~12 s, with and without its GroupNorm + SiLU staging
# transform (diagnostic build -DHSIDM_V2_ABLATE, HSIDM_V2_ABL=2 drops the transform: results are wrong, timing only), while rocm-smi samples
# shader clock and socket power once a second.  If the kernel were bound by instruction issue at a fixed clock, removing a third of its
# VALU work would shorten it at the SAME clock and LOWER power; if it is power-limited, the clock rises instead.
#   bash tools/power_ab.sh   (on the GPU box, from the repo root; writes gpurun_out/power_ab.txt)
out=gpurun_out/power_ab.txt
HSIDM_EXTRA_FLAGS="-DHSIDM_V2_ABLATE" HSIDM_OUT=$PWD/gpurun_out/libabl.so HSIDM_OBJ=/tmp/obj_abl bash hsi-dmgasr_amd/csrc/build.sh > /dev/null 2>&1
export HSIDM_LIB=$PWD/gpurun_out/libabl.so
: > $out
for shape in l32_768_256 l64_384_128; do
  for abl in 0 2 0 2; do
    echo "== $shape HSIDM_V2_ABL=$abl (0: product kernel, 2: staging transform removed)" >> $out
    HSIDM_V2_ABL=$abl python tools/conv_bench.py --batch 240 --reps 3 --only $shape --sustain 12 2>/dev/null | grep -v amdgpu >> $out &
    pid=$!
    sleep 5
    for i in 1 2 3 4 5; do
      rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics Package Power|Average Graphics Package Power" | sed 's/^/    /' >> $out
      sleep 1
    done
    wait $pid
  done
done
rm -f gpurun_out/libabl.so
cat $out
