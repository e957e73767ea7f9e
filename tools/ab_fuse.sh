#!/bin/bash
# Same-box A/B of the training view with and without the expansion fused into the rasterizer's per-Gaussian stages
# (CGS_FUSE_VIEW=0/1): views/s, ms per step and the kernel groups involved.  -> gpurun_out/ab_fuse.txt
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps"
mkdir -p gpurun_out
for rep in 1 2 3 4; do
  for f in 0 1; do
    CGS_FUSE_VIEW=$f python bench.py $F 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); k=j['kernels']
print('fuse=$f rep=$rep', j['value'], 'views/s', j['ms_per_step'], 'ms |', ' '.join('%s %.0f' % (n, k[n]['avg_us']) for n in ('preprocess','preprocess_bwd','expand_fwd','expand_bwd') if n in k))"
  done
done | tee gpurun_out/ab_fuse.txt
