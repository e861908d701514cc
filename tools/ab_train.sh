#!/bin/bash
# In-box A/B of an environment switch on the training step (bench.py --workload train, B = 4, bf16 set, graph replays):
#   bash tools/ab_train.sh HSIDM_TRAIN_WGRAD_STREAM=0
rounds=${ROUNDS:-3}
for r in $(seq $rounds); do
  for v in "" "$@"; do
    if [ -z "$v" ]; then tag=default; run() { python bench.py --workload train --steps ${STEPS:-100} 2>/dev/null; }
    else tag=$v; run() { env "$v" python bench.py --workload train --steps ${STEPS:-100} 2>/dev/null; }; fi
    run | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$tag', round(d['ms_per_step'],3), 'ms/step', round(d['value'],1))"
  done
done
