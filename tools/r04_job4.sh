#!/bin/bash
# round 4, GPU call 4: intermediate hi+lo policies between cout<=64 and cout<=128, each on every chain fixture + its step time
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
i=0
for pol in "cout <= 64 or (cout <= 128 and cin <= 128)" "cout <= 64 or (cout <= 128 and cin <= 256)" "cout <= 64 or (cout <= 128 and cin >= 192)"; do
  i=$((i+1))
  rm -f gpurun_out/parity.jsonl
  HSIDM_WIDE_POLICY="$pol" python -m pytest tests/test_gpu_chain.py -q -k "(T1000 or T20_chain_against) and fp16" --timeout 3000 > gpurun_out/job4_chain_p$i.log 2>&1
  tail -2 gpurun_out/job4_chain_p$i.log
  cp gpurun_out/parity.jsonl gpurun_out/job4_parity_p$i.jsonl
  HSIDM_WIDE_POLICY="$pol" python bench.py --steps 200 --no-modes --no-gae --no-train --no-small --no-parity --no-cpu-baseline --no-roofline > gpurun_out/job4_bench_p$i.json 2> gpurun_out/job4_bench_p$i.err
  echo "policy $i: $pol"; cut -c1-260 gpurun_out/job4_bench_p$i.json
done
