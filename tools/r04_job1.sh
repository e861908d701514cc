#!/bin/bash
# round 4, GPU call 1: where the round-3 build stands on the new chain fixture set + SQ counters of the two dominant fp16 kernels
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_chain.py -q -k "T20_chain" -x --timeout 1500 > gpurun_out/job1_chain.log 2>&1
tail -5 gpurun_out/job1_chain.log
bash tools/pmc_conv.sh l128_64_64 240 fp16 > gpurun_out/job1_pmc_v3.log 2>&1
bash tools/pmc_conv.sh l64_128_128 240 fp16 > gpurun_out/job1_pmc_v2.log 2>&1
tail -40 gpurun_out/job1_pmc_v3.log
