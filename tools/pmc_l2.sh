#!/bin/bash
# Run ON THE GPU BOX: the L2 <-> fabric counters of the C2 step under three store policies
# (st_plain = 0: every row streamed, 4: the shipped mix, 16: every row write-back).
# Usage: GIT_SHA=<sha> tools/pmc_l2.sh <tag>  -> gpurun_out/profile_<tag>sp{0,4,16}_c2/summary.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
for sp in 4 0 16; do
  SMX_PROFILE_EXTRA=1 SMX_PROFILE_ONLY_L2=1 SMX_PROFILE_PASS_S=120 \
  SMX_PROFILE_CMD="python3 $R/bench.py --config c2 --no-cpu-baseline --no-other-configs --steps 30 --warmup 10 --steps-per-graph 10 --opts st_plain=$sp" \
    bash "$R/tools/collect_profile.sh" "${TAG}sp$sp" c2 > /dev/null 2>&1
  echo "st_plain=$sp done" >> "$R/gpurun_out/pmc_l2_progress.txt"
done
python3 - "$R" "$TAG" <<'PY'
import json, sys
r, tag = sys.argv[1:3]
names = None
for sp in (0, 4, 16):
    d = json.load(open(f"{r}/gpurun_out/profile_{tag}sp{sp}_c2/summary.json"))
    ks = {k["name"]: k["avg_us"] for k in d.get("kernel_stats", [])}
    for k, v in d["counters_per_launch"].items():
        if "k_fused" not in k: continue
        print(f"st_plain={sp:2d} {k[5:18]} stats-pass avg {ks.get(k, 0):6.1f} us | " + " ".join(f"{c.replace('TCC_','').replace('_sum','')}={x:.3g}" for c, x in sorted(v.items())))
PY
