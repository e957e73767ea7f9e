#!/bin/bash
# rows per workgroup of wgrad_multi_kernel (the level MLPs' weight gradients on the 15 % rate subset: ~60 k rows per level)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps --steps 60"
for rep in 1 2; do
 for v in ${VARIANTS:-product wgm512 wgm1024 wgm2048}; do
  if [ $v = product ]; then E="X=1"; else E="CGS_LIB_PATH=tools/variants/libcgs_$v.so CGS_LIB_ALLOW_STALE=1"; fi
  env $E timeout -k 5 300 python bench.py $F 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); k=j['kernels']
print('%-8s rep=$rep' % '$v', j['value'], 'views/s', j['ms_per_step'], 'ms | level_mlp_wgrad %.1f us x%d | mlp_wgrad %.1f us | hip kernels %s' % (k['level_mlp_wgrad']['avg_us'], k['level_mlp_wgrad']['launches']//j['steps'], k.get('mlp_wgrad',{}).get('avg_us',0), j.get('hip_kernel_ms_per_step')))"
 done
done | tee gpurun_out/wgm_ab.txt
