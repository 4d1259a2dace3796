#!/bin/bash
# After tools/collect_profile.sh r03 {c2,c3,c5,f2,f2b8} and tests/tools/conv_bench.py > gpurun_out/r03p_next_rows_bench.txt
# on the GPU box: copy the condensed files into profiles/ under their tracked names.
set -euo pipefail
cd "$(dirname "$0")/.."
for c in c2 c3 c5; do
  cp gpurun_out/profile_r03_$c/summary.json profiles/r03_${c}_summary.json
  cp gpurun_out/profile_r03_$c/kernel_stats.csv profiles/r03_${c}_kernel_stats.csv
  cp gpurun_out/profile_r03_$c/bench_line.json profiles/r03_${c}_bench_line_under_rocprof.json
done
cp gpurun_out/profile_r03_f2/summary.json profiles/r03_f2_conv1_summary.json
cp gpurun_out/profile_r03_f2/kernel_stats.csv profiles/r03_f2_conv1_kernel_stats.csv
cp gpurun_out/profile_r03_f2b8/summary.json profiles/r03_f2_conv1_b8_summary.json
cp gpurun_out/profile_r03_f2b8/kernel_stats.csv profiles/r03_f2_conv1_b8_kernel_stats.csv
{ echo "# python tests/tools/conv_bench.py (defaults), git $(git rev-parse --short HEAD), one box; final build of round 3 (k_conv1 on 16- / 32-channel workgroups, smx_conv_response, smx_phase_filter, k_conv_grads, folded rows, cached Hermitian scale, tile counts 17...31 and 36...240 on the four-step path); profiles/r03_next_rows_bench_session1.txt is the same command at the end of the round's first session"
  grep '"op"' gpurun_out/r03p_next_rows_bench.txt; } > profiles/r03_next_rows_bench.txt
python3 - <<'PY'
import json
for c in ('c2', 'c3', 'c5'):
    s = json.load(open(f'profiles/r03_{c}_summary.json')); b = json.load(open(f'profiles/r03_{c}_bench_line_under_rocprof.json'))
    print(c, s['git_sha'], s['libsmx_sha256'][:12], 'ms', b['ms_per_step'], 'frac', b['hbm_roofline_frac_fwd_bwd'])
for f in ('profiles/r03_f2_conv1_summary.json', 'profiles/r03_f2_conv1_b8_summary.json'):
    s = json.load(open(f))
    print(f, [(k['name'][:34], round(k['avg_us'], 1)) for k in s['kernel_stats'][:6]])
PY
