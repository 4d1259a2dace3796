#!/bin/bash
# After tools/collect_all.sh <tag> on the GPU box: copy the condensed files into profiles/ under their tracked names.
#   tools/store_profiles.sh <collected tag> <name in profiles/>      e.g.  tools/store_profiles.sh r04a r04
set -euo pipefail
cd "$(dirname "$0")/.."
T=${1:-r04}; N=${2:-r04}
for c in c2 c3 c5 f2 block; do
  [ -f gpurun_out/profile_${T}_$c/summary.json ] || { echo "no profile for $c"; continue; }
  cp gpurun_out/profile_${T}_$c/summary.json profiles/${N}_${c}_summary.json
  cp gpurun_out/profile_${T}_$c/kernel_stats.csv profiles/${N}_${c}_kernel_stats.csv
  [ -s gpurun_out/profile_${T}_$c/bench_line.json ] && cp gpurun_out/profile_${T}_$c/bench_line.json profiles/${N}_${c}_bench_line_under_rocprof.json
done
[ -f gpurun_out/${T}_block_bench.txt ] && grep -h '^{' gpurun_out/${T}_block_bench.txt > profiles/${N}_block_bench.txt || true
python3 - "$N" <<'PY'
import json, sys
n = sys.argv[1]
for c in ('c2', 'c3', 'c5'):
    s = json.load(open(f'profiles/{n}_{c}_summary.json')); b = json.load(open(f'profiles/{n}_{c}_bench_line_under_rocprof.json'))
    print(c, s['git_sha'], s['libsmx_sha256'][:12], 'ms', b['ms_per_step'], 'frac', b['hbm_roofline_frac_fwd_bwd'],
          [(k['name'][:28], round(k['avg_us'], 1)) for k in s['kernel_stats'][:4]])
PY
