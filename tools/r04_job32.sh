#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/j32
python -m pytest tests/test_gpu_chain.py -x -q -k "T1000" > gpurun_out/j32/t1000.log 2>&1; tail -3 gpurun_out/j32/t1000.log
grep "T1000" gpurun_out/parity.jsonl | tail -8 | cut -c1-330
