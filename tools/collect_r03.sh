#!/bin/bash
# Round-3 evidence (run through gpurun from the repo root): bash tools/collect_r03.sh   -> gpurun_out/r03/
#   attention_v2 SQ counters (MFMA utilisation, north_star: "MFMA utilisation on attention against chip peak"), per precision mode
#   kernel_stats of the fp32 mode, of the bf16 mode and of the group autoencoder; FETCH / WRITE passes of the fp32 and fp16 modes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r03
mkdir -p $out
# ---- attention: three SQ passes at 240 images (bf16 and fp16 element types)
p1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS"
p2="SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAIT_INST_LDS"
i=0
for pass in "$p1" "$p2"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/att_pass$i -o p -- python tools/attn_bench.py --batch 240 --reps 2 > $out/att_pass$i.log 2>&1
  f=$(find $out/att_pass$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" $out/attention_b240_pass$i.csv
  rm -rf $out/att_pass$i
done
python tools/attn_bench.py --batch 240 > $out/attention_b240_time.txt 2>&1
# ---- per-kernel tables of the other two modes and of the autoencoder
for mode in fp32 bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$mode -o s -- python bench.py --precision $mode --steps 30 --warmup 5 --no-cpu-baseline --no-modes --no-parity --no-gae --no-train --no-small > $out/stats_$mode.log 2>&1
  cp $(find $out/st_$mode -name "*kernel_stats.csv" | head -1) $out/kernel_stats_$mode.csv
  rm -rf $out/st_$mode
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_gae -o s -- python tools/gae_bench.py > $out/gae_bench.txt 2>&1
cp $(find $out/st_gae -name "*kernel_stats.csv" | head -1) $out/kernel_stats_gae.csv
rm -rf $out/st_gae
# ---- HBM-side traffic of the fp32 mode's and the fp16 mode's conv kernels (separate FETCH / WRITE passes, eager launches)
for mode in fp32 fp16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    HSIDM_NO_GRAPH=1 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${mode}_$c -o p -- python bench.py --precision $mode --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-modes --no-parity --no-gae --no-train --no-small > $out/pmc_${mode}_$c.log 2>&1
    cp $(find $out/pmc_${mode}_$c -name "*counter_collection.csv" | head -1) $out/${mode}_$c.csv
    rm -rf $out/pmc_${mode}_$c
  done
  python tools/hbm_traffic.py $out/${mode}_FETCH_SIZE.csv $out/${mode}_WRITE_SIZE.csv 240 > $out/hbm_traffic_$mode.json 2> $out/hbm_$mode.err
done
ls -la $out
