#!/bin/bash
# round 6, final evidence (run through gpurun from the repo root): the default bench line + detail legs + per-kernel table + HBM traffic
# (tools/collect_profiles.sh), the driver-form line, SQ counters of the four largest conv instances of the step (one-pass fp16 sets; the 64 -> 64
# conv_v3 launch with and without its fused projection), the attention counters, the N > 1 code path under a one-rank RCCL group, smoke.
# The -m gpu suite's log / parity.jsonl of the same tree are collected by a separate call.
cd "$GRAFT_REPO_ROOT"
bash tools/collect_profiles.sh r06_final detail > gpurun_out/r06_final_collect.log 2>&1
tail -c 600 gpurun_out/r06_final/bench.json
python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out gpurun_out/r06_final/driver_form_detail.json > gpurun_out/r06_final/bench_driver_form.json 2> gpurun_out/r06_final/bench_driver_form.err
tail -1 gpurun_out/r06_final/bench_driver_form.json | cut -c1-400
export HSIDM_EXPERIMENTAL=1
for s in l32_256_256 l128_64_64 l128_64_64_proj128 l64_128_128; do bash tools/pmc_conv.sh $s 240 fp16x1 > /dev/null 2>&1; done
mkdir -p gpurun_out/r06_pmc; cp -r gpurun_out/pmc_l32_256_256_fp16x1 gpurun_out/pmc_l128_64_64_fp16x1 gpurun_out/pmc_l128_64_64_proj128_fp16x1 gpurun_out/pmc_l64_128_128_fp16x1 gpurun_out/r06_pmc/ 2>/dev/null
find gpurun_out/r06_pmc -name "*.csv" -delete; find gpurun_out/r06_pmc -type d -name "pass*" -exec rm -rf {} + 2>/dev/null
bash tools/attn_pmc.sh r06_attention > /dev/null 2>&1
HSIDM_FORCE_DIST=1 python bench.py --gpus 1 --total-patches 64 --steps 100 --no-cpu-baseline --no-parity --detail-out gpurun_out/r06_final/force_dist_strong_detail.json > gpurun_out/r06_final/bench_force_dist_strong.json 2> gpurun_out/r06_final/bench_force_dist_strong.err
tail -1 gpurun_out/r06_final/bench_force_dist_strong.json | cut -c1-300
HSIDM_FORCE_DIST=1 python bench.py --gpus 1 --workload train --steps 30 > gpurun_out/r06_final/bench_force_dist_train.json 2> gpurun_out/r06_final/bench_force_dist_train.err
tail -1 gpurun_out/r06_final/bench_force_dist_train.json | cut -c1-300
python __graft_entry__.py smoke > gpurun_out/r06_final/smoke.log 2>&1; tail -5 gpurun_out/r06_final/smoke.log
