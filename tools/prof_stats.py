#!/usr/bin/env python3
"""Print the per-kernel table of a rocprofv3 --kernel-trace --stats run:  python tools/prof_stats.py DIR [steps]"""
import csv
import glob
import sys

d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%6.2f%% %9.1f us avg %7d calls  %s" % (100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3, int(r["Calls"]), r["Name"][:120]))
print("kernel time per step: %.3f ms over %g steps" % (tot / 1e6 / steps, steps))
