#!/bin/bash
# A/B of library builds on the SURVEY 8(f) rows (tests/tools/conv_bench.py, hipGraph figures), fresh process per candidate,
# interleaved rounds:  tools/ab_rows.sh libsmx_prev.so libsmx.so
C=$(cd "$(dirname "$0")/.." && pwd)
ROUNDS=${ROUNDS:-2}
for round in $(seq $ROUNDS); do
  for lib in "$@"; do
    SMX_LIB=$C/tensor-cuda-fft-_amd/csrc/$lib timeout -k 10 300 python3 "$C/tests/tools/conv_bench.py" --no-torch ${ROWS_ARGS:---pair none} 2>/dev/null | grep '"op"' | python3 -c "
import sys, json
out = []
for l in sys.stdin:
    d = json.loads(l); out.append('%s %s %.4f' % (d['op'][:10].replace(' ', '_'), d['shape'].split()[0], d.get('graph_ms') or d['ms']))
print('$lib round $round | ' + ' | '.join(out), flush=True)
"
  done
done
