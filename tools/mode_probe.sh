#!/bin/bash
# The step-time "mode" of a process against its in-step kernel durations: N fresh processes of the bench under
# rocprofv3 --kernel-trace --stats; prints step time and the average duration of every smx kernel of that process.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-6}
cd /tmp && export TMPDIR=/tmp SMX_BENCH_NO_BOX=1
for i in $(seq $N); do
  O=$R/gpurun_out/mode_probe/$i
  rm -rf "$O"; mkdir -p "$O"
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -- python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > "$O/log.txt" 2>&1
  python3 - "$O" <<'PY'
import sys, json, glob, csv
o = sys.argv[1]
line = [l for l in open(o + "/log.txt") if l.startswith("{")]
d = json.loads(line[-1]) if line else {}
ks = glob.glob(o + "/*/*kernel_stats.csv")
rows = [r for r in csv.DictReader(open(ks[0])) if "smx::" in r["Name"]] if ks else []
print("step %.4f | " % d.get("ms_per_step", -1) + " | ".join("%s n=%s avg %.1f min %.1f max %.1f" % (r["Name"].split("smx::")[1][:14], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3) for r in rows[:3]), flush=True)
PY
done
