#!/bin/bash
# L2-side traffic of ONE tools/conv_bench.py shape from the FETCH_SIZE / WRITE_SIZE passes (gfx950: bytes = (2 * FETCH + WRITE) * 1024):
#   bash tools/pmc_traffic_conv.sh SHAPE [BATCH] [PRECISION]        (environment switches such as HSIDM_NO_XCD_MAP=1 pass through)
shape=$1; batch=${2:-240}; prec=${3:-fp16x1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmct_$shape; rm -rf $out; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -o p -- python tools/conv_bench.py --batch $batch --reps 2 --precision $prec --only $shape > $out/$c.log 2>&1
done
python - "$out" <<'P'
import csv, glob, sys
out = sys.argv[1]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + "/%s/**/*counter_collection.csv" % c, recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == c and "conv_v" in r["Kernel_Name"]]
    tot[c] = sum(v) / max(len(v), 1)
print("%s: FETCH %.1f MB (x2 = %.1f) WRITE %.1f MB -> %.1f MB per launch" % (out, tot["FETCH_SIZE"] / 1024, 2 * tot["FETCH_SIZE"] / 1024, tot["WRITE_SIZE"] / 1024, (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / 1024))
P
rm -rf $out/FETCH_SIZE $out/WRITE_SIZE
