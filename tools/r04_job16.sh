#!/bin/bash
# round 4, GPU call 16: the folded up / down-sampling forms with the sparse second weight pass: correctness, per-launch and whole-step A/B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/parity.jsonl
timeout 1200 python -m pytest tests/test_gpu_anchor.py tests/test_gpu_unet.py -q -x --timeout 600 2>&1 | tail -4
for shape in up_128_to128 down_128_64 down_64_128; do
  echo "sparse lo:"; python tools/conv_bench.py --batch 240 --reps 5 --precision fp16 --only $shape --sustain 2
  echo "dense lo:";  HSIDM_NO_SPARSE_LO=1 python tools/conv_bench.py --batch 240 --reps 5 --precision fp16 --only $shape --sustain 2
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/job16_conv_bench.txt
timeout 900 python -m pytest tests/test_gpu_chain.py -q -k "(T1000 or T20_chain_against) and fp16" --timeout 3000 2>&1 | tail -3
STEPS=300 ROUNDS=2 bash tools/ab_env_light.sh HSIDM_NO_SPARSE_LO=1 2>&1 | tee gpurun_out/job16_ab.txt
