#!/bin/bash
# anchor-MLP forward through raw buffer accesses (product) against the predicated global accesses of rounds 1-4
# (tools/variant_lib.sh m3fwd_old mlp3.hip -DM3_FWD_BUF=0): MLP / expansion tests, then interleaved short bench lines on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 5 900 python -m pytest tests/test_mlp_gpu.py tests/test_anchor_gen_gpu.py tests/test_context_gpu.py tests/test_training_gpu.py tests/test_edge_cases_gpu.py -x -q 2>&1 | tail -3 | tee gpurun_out/m3fwd_tests.txt
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps --steps 60"
for rep in 1 2 3; do
 for v in old new; do
  if [ $v = new ]; then E="X=1"; else E="CGS_LIB_PATH=tools/variants/libcgs_m3fwd_old.so CGS_LIB_ALLOW_STALE=1"; fi
  env $E timeout -k 5 300 python bench.py $F 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); k=j['kernels']
print('$v rep=$rep', j['value'], 'views/s', j['ms_per_step'], 'ms | mlp_fwd %.0f us x%d, mlp_bwd %.0f us' % (k['mlp_fwd']['avg_us'], k['mlp_fwd']['launches']//j['steps'], k['mlp_bwd']['avg_us']), '| hip kernels', j.get('hip_kernel_ms_per_step'))"
 done
done | tee gpurun_out/m3fwd_ab.txt
