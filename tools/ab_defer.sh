#!/bin/bash
# Same-box A/B: weight gradients launched inline (default on one GPU) vs deferred to the end of the backward
# (CGS_DEFER_WGRAD=1; what dist.GradientSync switches on for world > 1).  Headline step only.
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps"
mkdir -p gpurun_out
for rep in 1 2; do
  for d in 0 1; do
    CGS_DEFER_WGRAD=$d python bench.py $F 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); print('defer=$d rep=$rep', j['value'], 'views/s', j['ms_per_step'], 'ms', 'wgrad avg us', j['kernels']['mlp_wgrad']['avg_us'])"
  done
done | tee gpurun_out/ab_defer.txt
