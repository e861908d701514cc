#!/bin/bash
# per-kernel view of the fused-projection A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/j23; mkdir -p $out
for v in fused split; do
  if [ $v = split ]; then export HSIDM_NO_FUSED_PROJ=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/st_$v -o s -- python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline --no-modes --no-parity --no-gae --no-train --no-small > $out/$v.log 2>&1
  python tools/prof_stats.py $out/st_$v 70 16 > $out/$v.txt
  rm -rf $out/st_$v
done
