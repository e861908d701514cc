#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_final
python bench.py > gpurun_out/r04_final/bench.json 2> gpurun_out/r04_final/bench.err
tail -1 gpurun_out/r04_final/bench.json | cut -c1-300
python -m pytest tests/test_gpu_chain.py -q -k "bench" > gpurun_out/r04_final/bench_line_test.log 2>&1; tail -2 gpurun_out/r04_final/bench_line_test.log
