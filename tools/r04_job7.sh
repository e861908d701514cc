#!/bin/bash
# round 4, GPU call 7: conv_v2's 128-cout GN+SiLU form with the sparse second weight pass: correctness and time against the dense form
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/parity.jsonl
timeout 900 python -m pytest tests/test_gpu_anchor.py tests/test_gpu_unet.py -q -x --timeout 600 > gpurun_out/job7_tests.log 2>&1
tail -4 gpurun_out/job7_tests.log
for shape in l64_128_128 l64_384_128; do
  echo "sparse lo:"; python tools/conv_bench.py --batch 240 --reps 5 --precision fp16 --only $shape --sustain 3
  echo "dense lo:";  HSIDM_NO_SPARSE_LO=1 python tools/conv_bench.py --batch 240 --reps 5 --precision fp16 --only $shape --sustain 3
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/job7_conv_bench.txt
timeout 900 python -m pytest tests/test_gpu_chain.py -q -k "(T1000 or T20_chain_against) and fp16" --timeout 3000 > gpurun_out/job7_chain.log 2>&1
tail -3 gpurun_out/job7_chain.log
python bench.py --steps 300 --no-modes --no-gae --no-train --no-small --no-parity --no-cpu-baseline > gpurun_out/job7_bench.json 2> gpurun_out/job7_bench.err
cut -c1-330 gpurun_out/job7_bench.json
