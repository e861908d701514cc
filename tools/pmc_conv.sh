#!/bin/bash
# SQ counter passes over one tools/conv_bench.py shape:  bash tools/pmc_conv.sh SHAPE [BATCH] [PRECISION]   (writes gpurun_out/pmc_SHAPE_PRECISION/)
shape=$1; batch=${2:-120}; prec=${3:-bf16}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_${shape}_$prec
mkdir -p $out
p1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"
p2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
p3="SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INST_LEVEL_LDS"
i=0
for pass in "$p1" "$p2" "$p3"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $out/pass$i -o p -- python tools/conv_bench.py --batch $batch --reps 1 --precision $prec --only $shape > $out.log 2>&1
  f=$(find $out/pass$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "$f" | grep -A12 "conv_v\|conv1x1\|conv<" | tee $out/pass$i.txt
done
