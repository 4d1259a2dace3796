for rd in 512 0 512 0; do
python tools/kbench.py --shapes 64x4096x512x128,64x4096x512x256,128x4096x256x128,8x65536x256x128,64x4096x1024x512 --iters 20 --opts "round=$rd" 2>/dev/null | grep shape | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('round=$rd', d['shape'], 'fwd %.1f/%.1f bwd %.1f/%.1f frac %.3f'%(d['fwd_ms']*1e3,d['fwd_min']*1e3,d['bwd_ms']*1e3,d['bwd_min']*1e3,d['roofline_frac']))
"
done
