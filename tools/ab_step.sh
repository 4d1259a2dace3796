#!/bin/bash
# Step-level A/B: the driver's bench protocol (hipGraph of K fwd+bwd steps) for several builds of the library, each in
# a fresh process, interleaved over ROUNDS rounds:  tools/ab_step.sh libsmx_r03.so libsmx.so ...
# (store / load cache policies only show at this level: a launch timed alone leaves its dirty lines to the next one)
# BENCH_ARGS="--batch 64 --seq 4096 --dim 512 --filters 128" times another shape (no other_configs then)
C=$(cd "$(dirname "$0")/.." && pwd)/tensor-cuda-fft-_amd/csrc
ROUNDS=${ROUNDS:-2}
# A candidate is "lib.so" or "lib.so:knob=value;knob=value" (bench.py --opts)
for round in $(seq $ROUNDS); do
  for cand in "$@"; do
    lib=${cand%%:*}; opts=""; [ "$cand" != "$lib" ] && opts=${cand#*:}
    SMX_LIB=$C/$lib timeout -k 10 180 python3 "$C/../../bench.py" --steps 20 --warmup 5 --no-cpu-baseline --opts "$opts" ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); o = d.get('other_configs', {})
    f = lambda k: ('%.4f' % o[k]['ms_per_step']) if k in o and 'ms_per_step' in o[k] else 'n/a'
    L = d['roofline']['launches']
    print('$cand round $round  c2 %.4f ms (%.3f)  fwd %.1f bwd %.1f us | c3 %s c5 %s f2 %s' % (d['ms_per_step'], d['hbm_roofline_frac_fwd_bwd'], L['forward']['avg_ms']*1e3, L['backward']['avg_ms']*1e3, f('c3'), f('c5'), f('f2')), flush=True)
"
  done
done
