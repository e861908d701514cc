python -m pytest tests/test_gpu_unet.py tests/test_gpu_anchor.py -q -m gpu -x -k "conv or anchor or block or unet" 2>&1 | tail -5
for prec in bf16 fp16; do for sh in l128_64_64 l128_192_64; do python tools/conv_bench.py --batch 240 --precision $prec --only $sh --reps 8 2>&1 | tail -1; done; done
