#!/bin/bash
# A/B of library builds inside ONE gpurun call, interleaved rounds: tools/ab.sh libA.so libB.so ...
C=$PWD/tensor-cuda-fft-_amd/csrc
SHAPES=${SHAPES:-64x4096x256x128}
for round in 1 2 3; do
  for lib in "$@"; do
    SMX_LIB=$C/$lib timeout -k 10 120 python tools/kbench.py --shapes $SHAPES --iters 30 --opts "${OPTS:-}" 2>/dev/null | grep shape | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$lib round $round', d['shape'], 'fwd %.1f/%.1f bwd %.1f/%.1f spec %.1f us frac %.3f'%(d['fwd_ms']*1e3,d['fwd_min']*1e3,d['bwd_ms']*1e3,d['bwd_min']*1e3,d['spec_ms']*1e3,d['roofline_frac']))
"
  done
done
