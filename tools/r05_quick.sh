#!/bin/bash
# Round-5 quick GPU check: the context / level tests, then an A/B of the headline step with the fused level kernels on / off.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_ctx_level_gpu.py tests/test_training_parity_gpu.py tests/test_context_gpu.py tests/test_ctx_ops_gpu.py tests/test_training_gpu.py tests/test_edge_cases_gpu.py -x -q 2>&1 | tail -25 | tee gpurun_out/r05_quick_tests.txt
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps --steps 60"
for rep in 1 2; do
 for fused in 0 1; do
  CGS_LEVEL_FUSED=$fused timeout 300 python bench.py $F 2>gpurun_out/r05_quick_bench_err_$fused.txt | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); k=j['kernels']
print('fused=$fused rep=$rep', j['value'], 'views/s', j['ms_per_step'], 'ms |', ' '.join('%s %.0fus x%d' % (n, k[n]['avg_us'], k[n]['launches']//j['steps']) for n in ('mlp_fwd','mlp_bwd','mlp_wgrad','ctx_fwd','ctx_bwd','rate_fwd','rate_bwd') if n in k), '| hip kernels', j.get('hip_kernel_ms_per_step'))"
 done
done | tee gpurun_out/r05_quick_bench.txt
