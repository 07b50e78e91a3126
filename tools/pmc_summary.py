#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean of each counter over dispatches."""
import csv
import glob
import sys
from collections import defaultdict

for root in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(list))
    for fn in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            if "anonymous" not in k or "at::" in k:
                continue
            acc[k.split("(")[1 if k.startswith("(") else 0][:60] if False else k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k)
        for c, v in sorted(cs.items()):
            print(f"   {c:32s} {sum(v) / len(v):16.0f}  (n={len(v)})")
