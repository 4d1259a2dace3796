#!/bin/bash
# Run ON THE GPU BOX: the step-time "mode" of a process against hardware counters.
# N fresh processes of the default bench under rocprofv3 --pmc; each run's own dispatch timestamps give the
# kernel durations, so every run classifies itself (fast / slow) and carries its counters.
# Usage: tools/mode_pmc.sh <outdir> <runs> <counter> [<counter> ...]
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$1; N=$2; shift 2
mkdir -p "$O"
O=$(cd "$O" && pwd)
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $N); do
  rocprofv3 --pmc "$@" --output-format csv -d "$O/run$i" -- python3 $R/bench.py --no-cpu-baseline --no-other-configs --steps 50 --warmup 10 > "$O/run$i.log" 2>&1
  echo "run $i done: $(grep -h '^{' $O/run$i.log | tail -1 | cut -c1-120)"
done
python3 $R/tools/mode_pmc_summary.py "$O" > "$O/summary.json"
cat "$O/summary.json"
