#!/bin/bash
# In-box A/B of an environment switch of libhsidm.so on the headline benchmark:  STEPS=200 ROUNDS=3 bash tools/ab_env.sh HSIDM_NO_XCD_MAP=1
steps=${STEPS:-200}; rounds=${ROUNDS:-3}
for r in $(seq $rounds); do
  for v in "" "$@"; do
    if [ -z "$v" ]; then tag=default; run() { python bench.py --steps $steps --warmup 10 --no-cpu-baseline --no-parity --no-roofline 2>/dev/null; }
    else tag=$v; run() { env "$v" python bench.py --steps $steps --warmup 10 --no-cpu-baseline --no-parity --no-roofline 2>/dev/null; }; fi
    run | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); c=d.get('config',{}); print('$tag', round(c.get('ms_per_step_window') or d['ms_per_step'],3), 'ms/step (timed window)', round(c.get('value_window') or d['value'],1))"
  done
done
