#!/bin/bash
# round 4, GPU call 3: T=1000 reference chain, the cheaper hi+lo policy (cout <= 64) on every fixture, smoke, full GPU suite
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/parity.jsonl
python -m pytest tests/test_gpu_chain.py -q -k "T1000 or T20_chain_against" --timeout 3000 > gpurun_out/job3_chain.log 2>&1
tail -5 gpurun_out/job3_chain.log
cp gpurun_out/parity.jsonl gpurun_out/job3_parity_default.jsonl; rm -f gpurun_out/parity.jsonl
HSIDM_WIDE_POLICY="cout <= 64" python -m pytest tests/test_gpu_chain.py -q -k "(T1000 or T20_chain_against) and fp16" --timeout 3000 > gpurun_out/job3_chain_c64.log 2>&1
tail -5 gpurun_out/job3_chain_c64.log
cp gpurun_out/parity.jsonl gpurun_out/job3_parity_c64.jsonl; rm -f gpurun_out/parity.jsonl
HSIDM_WIDE_POLICY="cout <= 64" python bench.py --steps 300 --no-modes --no-gae --no-train --no-small --no-parity --no-cpu-baseline > gpurun_out/job3_bench_c64.json 2> gpurun_out/job3_bench_c64.err
cut -c1-330 gpurun_out/job3_bench_c64.json
python bench.py --steps 300 --no-modes --no-gae --no-train --no-small --no-parity --no-cpu-baseline > gpurun_out/job3_bench_c128.json 2> gpurun_out/job3_bench_c128.err
cut -c1-330 gpurun_out/job3_bench_c128.json
python __graft_entry__.py smoke > gpurun_out/job3_smoke.log 2>&1; tail -5 gpurun_out/job3_smoke.log
python -m pytest tests -q -m gpu --timeout 3000 -x > gpurun_out/job3_suite.log 2>&1
tail -8 gpurun_out/job3_suite.log
