#!/bin/bash
# Where does a 1x1 GEMM launch (conv1x1_g.hip) spend its time?  Builds the library with one piece of the kernel removed at a time
# (-DG1_ABL=n: 1 activation loads, 2 weight loads, 3 MFMAs, 4 A-fragment LDS reads, 5 barriers, 6 output stores, 7 whole epilogue, 8 statistics) and times the 16x16 / 32x32 level
# shapes at batch 240.  Run on the GPU box from the repo root:  bash tools/g1_ablate.sh
for n in ${ABLS:-1 2 3 4 5 6 7 8}; do
  HSIDM_EXTRA_FLAGS="-DG1_ABL=$n" HSIDM_OUT=$PWD/gpurun_out/libg1_$n.so HSIDM_OBJ=/tmp/obj_g1_$n bash hsi-dmgasr_amd/csrc/build.sh > /dev/null 2>&1 &
done
wait
for n in 0 ${ABLS:-1 2 3 4 5 6 7 8}; do
  if [ $n = 0 ]; then unset HSIDM_LIB; else export HSIDM_LIB=$PWD/gpurun_out/libg1_$n.so; fi
  echo "G1_ABL=$n"
  for s in qkv_16_512 proj16_1024_512 out16_512_512 proj32_768_256; do
    python tools/conv_bench.py --batch ${BATCH:-240} --reps 7 --only $s 2>/dev/null | grep -v amdgpu
  done
done
rm -f gpurun_out/libg1_*.so
