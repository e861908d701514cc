#!/bin/bash
# fused projection in conv_v3 (PROJ): correctness, then the in-box A/B against the two-launch form
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j22
timeout 900 python -m pytest tests/test_gpu_anchor.py -x -q -k "projection_fused or v3" > gpurun_out/j22/anchor.log 2>&1
tail -5 gpurun_out/j22/anchor.log
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q > gpurun_out/j22/unet.log 2>&1
tail -5 gpurun_out/j22/unet.log
STEPS=200 ROUNDS=3 bash tools/ab_env_light.sh HSIDM_NO_FUSED_PROJ=1 > gpurun_out/j22/ab.txt 2>&1
cat gpurun_out/j22/ab.txt
timeout 1200 python -m pytest tests/test_gpu_chain.py -x -q -k "T20 and fp16" > gpurun_out/j22/chain.log 2>&1
tail -5 gpurun_out/j22/chain.log
grep -h "chain\|proj" gpurun_out/parity.jsonl 2>/dev/null | tail -40 > gpurun_out/j22/parity_tail.txt
