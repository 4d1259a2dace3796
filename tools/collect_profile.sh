#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel stats + HBM counters of the bench command.
# Usage: tools/collect_profile.sh <tag>      -> gpurun_out/profile_<tag>/...
set -uo pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/profile_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# SMX_PROFILE_CMD overrides the profiled command (default: the bench), e.g. the block bench
CMD=${SMX_PROFILE_CMD:-"python3 $R/bench.py --no-cpu-baseline --steps 50 --warmup 10"}
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- $CMD > "$O/stats.log" 2>&1
# counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- $CMD > "$O/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- $CMD > "$O/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --output-format csv -d "$O/pmc_sq" -- $CMD > "$O/pmc_sq.log" 2>&1
python3 "$R/tools/summarize_profile.py" "$O" "$TAG" "$CMD" > "$O/summary.json"
cat "$O/summary.json"
