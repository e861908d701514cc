#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean counter values per dispatch."""
import csv
import collections
import sys

path = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    short = k.replace("_ZN5hsidm17conv_igemm_kernelINS_7ConvCfgI", "conv<")[:70]
    n = max(len(v) for v in cs.values())
    print("%s  (n=%d)" % (short, n))
    for c, v in sorted(cs.items()):
        print("    %-32s %16.1f" % (c, sum(v) / len(v)))
