#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from a build log made with
SMX_EXTRA="-Rpass-analysis=kernel-resource-usage" (tools/README.md)."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
only = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split(' ')[0]
    f = lambda k: re.search(k + r': (\d+)', b).group(1)
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0]
    dn = dn.replace('void ', '').replace('smx::', '').replace('(anonymous namespace)::', '')
    if only and not re.search(only, dn):
        continue
    scratch, occ, lds = f(r'ScratchSize \[bytes/lane\]'), f(r'Occupancy \[waves/SIMD\]'), f(r'LDS Size \[bytes/block\]')
    print(f"{dn:48s} vgpr {f('VGPRs'):>3} agpr {f('AGPRs'):>3} spill {f('VGPRs Spill'):>3} "
          f"scratch {scratch:>4} occ {occ} lds {lds}")
