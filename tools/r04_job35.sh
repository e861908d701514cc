#!/bin/bash
# SQ counters of the kernel instance that is now the largest of the step (conv_v2<256>: 32x32 and 16x16 levels) and of conv_v3's projection form, final build, 240 latents
cd "$GRAFT_REPO_ROOT"
for s in l32_256_256 l16_512_512 l128_64_64_proj128; do bash tools/pmc_conv.sh $s 240 fp16 > /dev/null 2>&1; done
python tools/conv_bench.py --batch 240 --precision fp16 --only l32_256_256 2>/dev/null | tail -3
python tools/conv_bench.py --batch 240 --precision fp16 --only l16_512_512 2>/dev/null | tail -3
python tools/conv_bench.py --batch 240 --precision fp16 --only l128_64_64_proj128 2>/dev/null | tail -3
ls gpurun_out | grep pmc_
