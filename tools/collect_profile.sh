#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel stats + HBM counters of the bench command.
# Usage: GIT_SHA=<sha> tools/collect_profile.sh <tag> [c2|c3|c5]   -> gpurun_out/profile_<tag>_<cfg>/...
# The summary is stamped with the git SHA handed in (the box has no .git) and the sha256 of the
# libsmx.so that was profiled; bench.py quotes roofline.traffic from it only for that same library.
set -uo pipefail
TAG=${1:-r02}
CFG=${2:-c2}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/profile_${TAG}_$CFG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# SMX_PROFILE_CMD overrides the profiled command (default: the bench), e.g. the block bench
# (10 steps per hipGraph: the counter passes serialise every kernel node, and a 50-step graph of the residue-split
#  plan -- 450 nodes per replay -- sat silent for seven minutes in the first --pmc pass of round 4)
CMD=${SMX_PROFILE_CMD:-"python3 $R/bench.py --config $CFG --no-cpu-baseline --no-other-configs --steps 50 --warmup 10 --steps-per-graph 10"}
T="timeout -k 10 ${SMX_PROFILE_PASS_S:-300}"
export SMX_BENCH_NO_BOX=1        # (the yardstick copies and the clock kernel are not part of the profiled step)
$T rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- $CMD > "$O/stats.log" 2>&1
# counters in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950)
echo "stats pass done" >> "$O/progress.txt"
if [ "${SMX_PROFILE_ONLY_L2:-0}" != "1" ]; then
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -- $CMD > "$O/pmc_fetch.log" 2>&1
echo "fetch pass done" >> "$O/progress.txt"
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -- $CMD > "$O/pmc_write.log" 2>&1
if [ "${SMX_PROFILE_SQ:-1}" = "1" ]; then
echo "write pass done" >> "$O/progress.txt"
$T rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
  --output-format csv -d "$O/pmc_sq" -- $CMD > "$O/pmc_sq.log" 2>&1
fi
fi
# optional extra passes (SMX_PROFILE_EXTRA=1; SMX_PROFILE_ONLY_L2=1 skips the passes above and the instruction mix): instruction mix, LDS / vector-memory issue, L2 <-> fabric requests
if [ "${SMX_PROFILE_EXTRA:-0}" = "1" ]; then
echo "sq pass done" >> "$O/progress.txt"
[ "${SMX_PROFILE_ONLY_L2:-0}" = "1" ] || $T rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR \
  --output-format csv -d "$O/pmc_x1" -- $CMD > "$O/pmc_x1.log" 2>&1
echo "x1 pass done" >> "$O/progress.txt"
# (the L2 counters are per channel: more than two or three of them "exceed the capabilities of the hardware" in one pass)
i=0
for pair in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
            "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
            "TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum" "TCC_STREAMING_REQ_sum TCC_REQ_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum"; do
  i=$((i+1))
  $T rocprofv3 --pmc $pair --output-format csv -d "$O/pmc_x2_$i" -- $CMD > "$O/pmc_x2_$i.log" 2>&1
  echo "x2 pass $i done ($pair)" >> "$O/progress.txt"
done
echo "x2 pass done" >> "$O/progress.txt"
fi
SHA=$(sha256sum "$R/tensor-cuda-fft-_amd/csrc/libsmx.so" | cut -d' ' -f1)
python3 "$R/tools/summarize_profile.py" "$O" "${TAG}_$CFG" "$CMD" "${GIT_SHA:-unknown}" "$SHA" > "$O/summary.json"
cp "$O"/stats/*/*kernel_stats.csv "$O/kernel_stats.csv" 2>/dev/null
grep -h '^{' "$O/stats.log" | tail -1 > "$O/bench_line.json"
# the raw rocprofv3 trees are tens of MiB (gpurun returns at most 64 MiB): keep the condensed files only
if [ "${SMX_PROFILE_KEEP_RAW:-0}" != "1" ]; then rm -rf "$O/stats" "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_sq" "$O/pmc_x1" "$O"/pmc_x2_*; fi
cat "$O/summary.json"
