#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --kernel-trace --stats run: tools/kstats.py <dir> [n]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in list(csv.DictReader(open(f)))[:n]:
    name = r["Name"].replace("void ", "").split("(")[0][:64]
    print("%-64s calls %5s avg %9.1f us  min %9.1f us  %5s %%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3,
                                                               float(r["MinNs"]) / 1e3, r["Percentage"]))
