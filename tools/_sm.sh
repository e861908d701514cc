cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for p in 1 8; do
out=gpurun_out/small_p$p; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python bench.py --patches $p --steps 200 --no-parity --no-modes --no-cpu-baseline --no-gae --no-small --no-train --no-roofline > $out/log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv; rm -rf $out/stats
grep '"metric"' $out/log | cut -c1-250
done
