#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-j28}; mkdir -p $out
shift
timeout 900 python -m pytest tests/test_gpu_anchor.py -x -q -k "projection_fused" > $out/anchor.log 2>&1
tail -3 $out/anchor.log
STEPS=200 ROUNDS=3 bash tools/ab_libs.sh "$@" > $out/ab_libs.txt 2>&1
cat $out/ab_libs.txt
HSIDM_PROBE_PREC=fp16 HSIDM_PROBE_PROJ=64,64 bash tools/v3_stamps.sh l128_64_64 > $out/stamps_proj_128.txt 2>&1
head -7 $out/stamps_proj_128.txt
