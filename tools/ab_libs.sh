#!/bin/bash
# In-box A/B of builds of libhsidm.so on the headline benchmark:  PREC=fp16 STEPS=200 ROUNDS=2 bash tools/ab_libs.sh OTHER.so [OTHER2.so ...]
# (box-to-box variance is +-4 %, so only numbers from the same box compare).  Prints ms/step per run, alternating with the default lib.
steps=${STEPS:-200}; rounds=${ROUNDS:-2}
for r in $(seq $rounds); do
  for lib in "" "$@"; do
    if [ -z "$lib" ]; then tag=base; unset HSIDM_LIB; else tag=$(basename $lib); export HSIDM_LIB=$PWD/$lib; fi
    python bench.py --precision ${PREC:-fp16} --steps $steps --warmup 10 --no-cpu-baseline --no-roofline --no-parity 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); c=d.get('config',{}); print('$tag', round(c.get('ms_per_step_window') or d['ms_per_step'],3), 'ms/step (timed window)', round(c.get('value_window') or d['value'],1))"
  done
done
