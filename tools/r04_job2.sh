#!/bin/bash
# round 4, GPU call 2: the precision schedule (first four steps of a chain in the fp32 mode) on the chain fixture set + a short bench
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/parity.jsonl
python -m pytest tests/test_gpu_chain.py -q -k "T20_chain or wrap" -x --timeout 1500 > gpurun_out/job2_chain.log 2>&1
tail -5 gpurun_out/job2_chain.log
HSIDM_BENCH_VERBOSE=1 python bench.py --steps 1000 --no-modes --no-gae --no-train --no-small --no-parity > gpurun_out/job2_bench.json 2> gpurun_out/job2_bench.err
tail -3 gpurun_out/job2_bench.err; cut -c1-600 gpurun_out/job2_bench.json
