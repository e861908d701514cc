#!/bin/bash
# Throughput vs latents per GPU (one process, N=1):  bash tools/batch_sweep.sh "4 8 12 16 20 24 32 48 64"
for p in ${1:-4 8 12 16 20 24 32 48 64}; do
  python bench.py --patches $p --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-fp32 --no-gae --no-train --no-small | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['batch_per_gpu'], round(d['value'],1), round(d['ms_per_step'],3))"
done
