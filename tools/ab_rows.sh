run() { env $1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-codec --no-image-loss 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$1', d['value'], d['value_raster_only'], k['blend_bwd']['avg_us'], k['blend_fwd']['avg_us'])"; }
for r in 1 2; do run CGS_BLEND_ROWS=0; run CGS_BLEND_ROWS=1; done
