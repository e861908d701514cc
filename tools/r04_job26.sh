#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/j26; mkdir -p $out
HSIDM_PROBE_PREC=fp16 HSIDM_PROBE_PROJ=64,64 bash tools/v3_stamps.sh l128_64_64 > $out/stamps_proj_128.txt 2>&1
HSIDM_PROBE_PREC=fp16 bash tools/v3_stamps.sh l128_64_64 > $out/stamps_plain.txt 2>&1
cat $out/stamps_proj_128.txt | head -20; head -6 $out/stamps_plain.txt
