#!/bin/bash
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-j24}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_anchor.py -x -q -k "projection_fused" > $out/anchor.log 2>&1
tail -3 $out/anchor.log
STEPS=200 ROUNDS=3 bash tools/ab_env_light.sh HSIDM_NO_FUSED_PROJ=1 > $out/ab.txt 2>&1
cat $out/ab.txt
