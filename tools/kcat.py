#!/usr/bin/env python3
"""Category totals of a rocprofv3 *_kernel_stats.csv: tools/kcat.py <csv> [calls per step]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
per = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
def cat(n):
    l = n.lower()
    if "smx::" in n or n.startswith("k_"): return "native (smx)"
    if n.startswith("Cijk_") or ("gemm" in l and "conv" not in l): return "GEMM (hipBLASLt / CK)"
    if "conv" in l or "im2" in l or "col2im" in l or "batched_transpose" in l: return "MIOpen / CK convolution + transposes"
    if "at::native" in n: return "torch elementwise / reduce / copy"
    return "other"
tot = collections.Counter(); calls = collections.Counter()
for r in rows:
    tot[cat(r["Name"])] += int(r["TotalDurationNs"]); calls[cat(r["Name"])] += int(r["Calls"])
al = sum(tot.values())
for k, v in tot.most_common():
    print("%-44s %8.2f ms per step  %5.1f %%  (%d launches per step)" % (k, v / 1e6 / per, 100.0 * v / al, calls[k] / per))
print("%-44s %8.2f ms per step" % ("all kernels", al / 1e6 / per))
top = sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[:14]
for r in top:
    print("   %7.2f ms  n=%-5s %s" % (int(r["TotalDurationNs"]) / 1e6 / per, r["Calls"], r["Name"][:110]))
