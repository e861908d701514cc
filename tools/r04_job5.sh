#!/bin/bash
# round 4, GPU call 5: per-kernel tables of the training step (bf16, B = 4) and of the group autoencoder at the bench batch (+ its HBM traffic)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r04
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_train -o s -- python bench.py --workload train --precision bf16 --steps 30 --warmup 5 > $out/train_bf16.log 2>&1
cp $(find $out/st_train -name "*kernel_stats.csv" | head -1) $out/kernel_stats_train_bf16_b4.csv; rm -rf $out/st_train
tail -1 $out/train_bf16.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_gae -o s -- python tools/gae_bench.py --cubes 48 --reps 10 > $out/gae_b48.txt 2>&1
cp $(find $out/st_gae -name "*kernel_stats.csv" | head -1) $out/kernel_stats_gae_b48_fp32.csv; rm -rf $out/st_gae
cat $out/gae_b48.txt | tail -1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_gae_$c -o p -- python tools/gae_bench.py --cubes 48 --reps 1 > $out/pmc_gae_$c.log 2>&1
  cp $(find $out/pmc_gae_$c -name "*counter_collection.csv" | head -1) $out/gae_b48_$c.csv; rm -rf $out/pmc_gae_$c
done
python tools/prof_stats.py $out 30 25 2>/dev/null | head -5
ls -la $out
