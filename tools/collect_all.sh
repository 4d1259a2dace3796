#!/bin/bash
# Run ON THE GPU BOX: every profile the round commits, one after the other, with a progress file that keeps gpurun's
# silence watchdog fed.  tools/collect_all.sh <tag> [git sha]   -> gpurun_out/profile_<tag>_{c2,c3,c5,f2,block}/
TAG=${1:-r04}
export GIT_SHA=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=$R/gpurun_out/collect_$TAG.progress
for c in c2 c3 c5; do
  echo "$(date +%T) $c" >> "$P"
  SMX_PROFILE_EXTRA=$([ $c = c2 ] && echo 1 || echo 0) "$R/tools/collect_profile.sh" $TAG $c > /dev/null 2>&1
done
echo "$(date +%T) f2" >> "$P"
SMX_PROFILE_CMD="python3 $R/tests/tools/conv_bench.py --full none --pair none --no-torch --conv 64x1024x512x128 --iters 20" \
  "$R/tools/collect_profile.sh" $TAG f2 > /dev/null 2>&1
echo "$(date +%T) block" >> "$P"
SMX_PROFILE_CMD="python3 $R/tools/block_bench.py --only fused_block" "$R/tools/collect_profile.sh" $TAG block > /dev/null 2>&1
python3 "$R/tools/block_bench.py" > "$R/gpurun_out/${TAG}_block_bench.txt" 2>&1
echo "$(date +%T) done" >> "$P"
for c in c2 c3 c5 f2 block; do echo "== $c"; cat "$R/gpurun_out/profile_${TAG}_$c/progress.txt" 2>/dev/null | tr '\n' ' '; echo; head -6 "$R/gpurun_out/profile_${TAG}_$c/kernel_stats.csv" | cut -c1-150; done
