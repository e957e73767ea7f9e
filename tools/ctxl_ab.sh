#!/bin/bash
# fused level kernels: product against a variant library (tools/variant_lib.sh <tag> ctx_level.hip ...): the level tests, the
# micro-benchmark of both, then interleaved short bench lines on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
V=${1:-clpipe0}
timeout -k 5 900 python -m pytest tests/test_ctx_level_gpu.py tests/test_training_parity_gpu.py tests/test_ctx_ops_gpu.py -x -q 2>&1 | tail -2
echo "product:"; timeout -k 5 200 python tools/ctxl_micro.py 800000 20 71 2>&1 | grep "us  "
echo "$V:"; CGS_LIB_PATH=tools/variants/libcgs_$V.so CGS_LIB_ALLOW_STALE=1 timeout -k 5 200 python tools/ctxl_micro.py 800000 20 71 2>&1 | grep "us  "
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps --steps 60"
for rep in 1 2; do
 for v in product $V; do
  if [ $v = product ]; then E="X=1"; else E="CGS_LIB_PATH=tools/variants/libcgs_$v.so CGS_LIB_ALLOW_STALE=1"; fi
  env $E timeout -k 5 300 python bench.py $F 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); k=j['kernels']
print('%-10s rep=$rep' % '$v', j['value'], 'views/s', j['ms_per_step'], 'ms | ctx_fwd %.0f us x%d, ctx_bwd %.0f us x%d | ctx group %s | hip kernels %s' % (k['ctx_fwd']['avg_us'], k['ctx_fwd']['launches']//j['steps'], k['ctx_bwd']['avg_us'], k['ctx_bwd']['launches']//j['steps'], j['ctx_group_roofline']['ms_per_step'], j.get('hip_kernel_ms_per_step')))"
 done
done | tee gpurun_out/ctxl_ab.txt
