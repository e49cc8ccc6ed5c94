#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 `*_kernel_stats.csv` (per-kernel calls / total / average / share)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print("%-64s calls %6s tot %9.2f ms avg %8.1f us %5.1f%%" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                              float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("total kernel time %.2f ms" % (tot / 1e6))
