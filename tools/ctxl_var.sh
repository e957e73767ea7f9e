#!/bin/bash
# the micro-benchmark of the level kernels on the product library and on every tools/variants/libcgs_$1*.so
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=${2:-800000}
( echo "product:"; python tools/ctxl_micro.py $N 20 71 2>&1 | grep "us  "
for lib in tools/variants/libcgs_$1*.so; do
  echo "$lib:"; CGS_LIB_PATH=$lib CGS_LIB_ALLOW_STALE=1 python tools/ctxl_micro.py $N 20 71 2>&1 | grep "us  "
done ) | tee gpurun_out/ctxl_var_$1.txt
