# nsplit / placement scan of the split path: SHAPE=BxNxDxF NS="4 8" PL="2" bash tools/c3scan.sh
for ns in ${NS:-8 16 32}; do for pl in ${PL:-2}; do
python tools/kbench.py --shapes ${SHAPE:-8x65536x256x128} --iters 15 --opts "nsplit=$ns;placement=$pl" 2>/dev/null | grep shape | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('ns=$ns pl=$pl', d['shape'], 'fwd %.1f/%.1f bwd %.1f/%.1f frac %.3f'%(d['fwd_ms']*1e3,d['fwd_min']*1e3,d['bwd_ms']*1e3,d['bwd_min']*1e3,d['roofline_frac']))
"
done; done
