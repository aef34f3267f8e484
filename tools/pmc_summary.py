#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel (per dispatch).
usage: pmc_summary.py DIR [DIR ...]"""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        per_dispatch = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            per_dispatch[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, k, c), v in per_dispatch.items():
            name = k.replace("(anonymous namespace)::", "").split("(")[0]
            agg[(name, c)].append(v)
        for (k, c), v in sorted(agg.items()):
            if v and max(v) > 0 and not k.startswith("__amd"):
                print(f"{k[:28]:28s} {c:38s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
