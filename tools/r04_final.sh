#!/bin/bash
# round 4, final evidence: full GPU suite (parity log), the default bench line + per-kernel table + HBM traffic, the N > 1 code path under a one-rank RCCL group
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04_final
rm -f gpurun_out/parity.jsonl
python -m pytest tests -q -m gpu --timeout 3000 > gpurun_out/r04_final/suite.log 2>&1
tail -3 gpurun_out/r04_final/suite.log
cp gpurun_out/parity.jsonl gpurun_out/r04_final/parity.jsonl
bash tools/collect_profiles.sh r04_final > gpurun_out/r04_final/collect.log 2>&1
HSIDM_FORCE_DIST=1 python bench.py --gpus 1 --total-patches 64 --steps 100 --no-cpu-baseline --no-modes --no-parity --no-gae --no-train --no-small > gpurun_out/r04_final/bench_force_dist_strong.json 2> gpurun_out/r04_final/bench_force_dist_strong.err
tail -1 gpurun_out/r04_final/bench_force_dist_strong.json | cut -c1-400
HSIDM_FORCE_DIST=1 python bench.py --gpus 1 --workload train --steps 30 > gpurun_out/r04_final/bench_force_dist_train.json 2> gpurun_out/r04_final/bench_force_dist_train.err
tail -1 gpurun_out/r04_final/bench_force_dist_train.json | cut -c1-400
python __graft_entry__.py smoke > gpurun_out/r04_final/smoke.log 2>&1; tail -4 gpurun_out/r04_final/smoke.log
