#!/bin/bash
# two short headline bench lines with the per-scope kernel averages (no tests)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps --steps 60"
for rep in 1 2; do
  timeout 300 python bench.py $F 2>gpurun_out/r05_bench_quick_err.txt | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); k=j['kernels']
print('rep=$rep', j['value'], 'views/s', j['ms_per_step'], 'ms |', ' '.join('%s %.0fus x%d' % (n, k[n]['avg_us'], k[n]['launches']//j['steps']) for n in ('mlp_fwd','mlp_bwd','mlp_wgrad','level_mlp_fwd','level_mlp_bwd','level_mlp_wgrad','ctx_fwd','ctx_bwd','rate_fwd','rate_bwd','expand_bwd','preprocess') if n in k), '| hip kernels', j.get('hip_kernel_ms_per_step'), '| ctx group', (j.get('ctx_group_roofline') or {}).get('ms_per_step'), (j.get('ctx_group_roofline') or {}).get('launches_per_step'))"
done | tee gpurun_out/r05_bench_quick.txt
tail -3 gpurun_out/r05_bench_quick_err.txt
