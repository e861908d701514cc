#!/bin/bash
# round 4, GPU call 8: in-box A/B of the sparse second weight pass on the whole step, then the full GPU suite
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
STEPS=300 ROUNDS=3 bash tools/ab_env_light.sh HSIDM_NO_SPARSE_LO=1 2>&1 | tee gpurun_out/job8_ab_sparse.txt
rm -f gpurun_out/parity.jsonl
python -m pytest tests -q -m gpu --timeout 3000 -x > gpurun_out/job8_suite.log 2>&1
tail -5 gpurun_out/job8_suite.log
