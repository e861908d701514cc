#!/bin/bash
# Regenerates the measurement artefacts of profiles/ on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh TAG [detail]   -> gpurun_out/TAG/{bench.json,bench_detail.json,kernel_stats.csv,fetch.csv,write.csv,hbm_traffic.json}
# bench.json = the compact contract line of `python bench.py` (defaults); "detail": bench.py --detail (the secondary legs in bench_detail.json).
# Copy hbm_traffic.json to profiles/hbm_traffic_fp16.json, which bench.py reads into roofline.traffic.
tag=${1:-final}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag
mkdir -p $out
python bench.py ${2:+--detail} --detail-out $out/bench_detail.json > $out/bench.json 2> $out/bench.err
# per-kernel table of the same command, 110 steps of which the chain's first eight run on the fp32 kernel set (rows with `Ef, 2` / f32x3)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity --detail-out $out/stats_detail.json > $out/stats.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  # (HSIDM_NO_STEP_SCHEDULE=1: the sampled steps are a chain's FIRST ones, which the fp16 policy runs on the fp32 kernel set - the
  # counters are wanted for the kernels of the other 992; HSIDM_NO_GRAPH=1: per-dispatch counters need eager launches)
  HSIDM_NO_STEP_SCHEDULE=1 HSIDM_NO_GRAPH=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-parity --detail-out $out/pmc_detail.json > $out/pmc_$c.log 2>&1
done
cp $(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $out/fetch.csv
cp $(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) $out/write.csv
batch=$(python -c "import json; print(json.loads(open('$out/bench.json').read().strip().splitlines()[-1])['config']['batch_per_gpu'])")
python tools/hbm_traffic.py $out/fetch.csv $out/write.csv $batch > $out/hbm_traffic.json 2> $out/hbm.err
rm -rf $out/stats $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/stats_detail.json $out/pmc_detail.json
cat $out/bench.json | cut -c1-600
head -30 $out/kernel_stats.csv | cut -c1-150
