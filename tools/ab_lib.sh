#!/bin/bash
# Same-box A/B of two builds of the library on the headline step: tools/ab_lib.sh <libA.so> <libB.so> [reps]
# ("" = the product build).  Prints views/s, ms per step and the per-group kernel averages bench.py reports.
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps"
A=$1; B=$2; reps=${3:-2}
mkdir -p gpurun_out
for rep in $(seq 1 $reps); do
  for L in "$A" "$B"; do
    CGS_LIB_ALLOW_STALE=1 CGS_LIB_PATH=$L python bench.py $F 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); k=j['kernels']
print('lib=${L:-product} rep=$rep', j['value'], 'views/s', j['ms_per_step'], 'ms |', ' '.join('%s %.0fx%d' % (n, k[n]['avg_us'], k[n]['launches']//j['steps']) for n in ('ctx_fwd','ctx_bwd','expand_bwd','rate_fwd','rate_bwd')))"
  done
done | tee gpurun_out/ab_lib.txt
