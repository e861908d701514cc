#!/bin/bash
# round 4, GPU call 11: SQ counters of the two sparse-second-pass kernels (the dense-pass counters of the same shapes: job 1), anchor tests
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_anchor.py -q -x --timeout 600 2>&1 | tail -3
rm -rf gpurun_out/pmc_l128_64_64_fp16 gpurun_out/pmc_l64_128_128_fp16
bash tools/pmc_conv.sh l128_64_64 240 fp16 > gpurun_out/job11_pmc_v3.log 2>&1
bash tools/pmc_conv.sh l64_128_128 240 fp16 > gpurun_out/job11_pmc_v2.log 2>&1
cat gpurun_out/pmc_l128_64_64_fp16/pass1.txt | head -12
