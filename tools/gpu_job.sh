#!/bin/bash
# Generic GPU-box job wrapper (run through gpurun from the repo root):  bash tools/gpu_job.sh NAME 'commands...'
# stdout/stderr of the commands go to gpurun_out/NAME.log; the tail is echoed so that gpurun's own tail shows it.
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( eval "$@" ) > gpurun_out/$name.log 2>&1
rc=$?
tail -c 3000 gpurun_out/$name.log
exit $rc
