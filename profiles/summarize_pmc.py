#!/usr/bin/env python3
"""Per-kernel mean of every counter in a rocprofv3 `*_counter_collection.csv` (one row per dispatch x counter)."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in acc.items():
    if pat and pat not in k:
        continue
    print(k, "dispatches", len(next(iter(cs.values()))))
    for c, v in sorted(cs.items()):
        print("    %-28s mean %.4g" % (c, sum(v) / len(v)))
