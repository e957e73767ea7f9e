#!/bin/bash
# Same-box A/B of an environment switch on the headline step: tools/ab_env.sh VAR [reps]   (VAR=0 vs VAR=1, alternating)
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps"
V=$1; reps=${2:-3}
mkdir -p gpurun_out
for rep in $(seq 1 $reps); do
  for f in 0 1; do
    env $V=$f python bench.py $F 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); k=j['kernels']
print('$V=$f rep=$rep', j['value'], 'views/s', j['ms_per_step'], 'ms | ctx_bwd %.0fx%d' % (k['ctx_bwd']['avg_us'], k['ctx_bwd']['launches']//j['steps']))"
  done
done | tee gpurun_out/ab_env_$V.txt
