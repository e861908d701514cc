#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-j25}; mkdir -p $out
shift
STEPS=200 ROUNDS=3 bash tools/ab_libs.sh "$@" > $out/ab_libs.txt 2>&1
HSIDM_NO_FUSED_PROJ=1 STEPS=200 ROUNDS=1 bash tools/ab_libs.sh >> $out/ab_libs.txt 2>&1
cat $out/ab_libs.txt
