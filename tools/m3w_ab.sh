#!/bin/bash
# anchor-MLP backward (mlp3_bwd_wg_kernel): product against a variant library, interleaved short bench lines on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
V=${1:-m3wpipe0}; V2=$2
timeout -k 5 600 python -m pytest tests/test_mlp_gpu.py tests/test_anchor_gen_gpu.py -x -q 2>&1 | tail -2
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps --steps 60"
for rep in 1 2; do
 for v in product $V $V2; do
  if [ $v = product ]; then E="X=1"; else E="CGS_LIB_PATH=tools/variants/libcgs_$v.so CGS_LIB_ALLOW_STALE=1"; fi
  env $E timeout -k 5 300 python bench.py $F 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); k=j['kernels']
print('%-10s rep=$rep' % '$v', j['value'], 'views/s', j['ms_per_step'], 'ms | mlp_bwd %.0f us, mlp_fwd %.0f us | hip kernels %s' % (k['mlp_bwd']['avg_us'], k['mlp_fwd']['avg_us'], j.get('hip_kernel_ms_per_step')))"
 done
done | tee gpurun_out/m3w_ab.txt
