#!/bin/bash
# SQ counters of the attention kernels at 240 images (north_star: "MFMA utilisation on attention against chip peak"), run through gpurun
# from the repo root:  bash tools/attn_pmc.sh TAG   -> gpurun_out/TAG/{attention_b240_pass1.csv,pass2.csv,attention_b240_time.txt,summary.txt}
# (HSIDM_ATTENTION_V1=2 in the environment: attention_v2, the form attention_v3 replaced)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-attention}
mkdir -p $out
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
python tools/attn_bench.py --batch 240 2>/dev/null > $out/attention_b240_time.txt
python tools/attn_util.py $out/attention_b240_pass1.csv $out/attention_b240_pass2.csv > $out/summary.txt 2>&1
cat $out/attention_b240_time.txt $out/summary.txt
