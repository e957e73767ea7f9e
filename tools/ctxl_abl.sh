#!/bin/bash
# timing ablations of the fused level kernels (tools/variants/libcgs_clabl<bits>.so, see CL_ABL in csrc/ctx_level.hip)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=${1:-800000}
( echo "product:"; python tools/ctxl_micro.py $N 20 71 2>&1 | grep "us  "
for ab in 1 2 4 8 16 3 31 32 64 128 224; do
  echo "CL_ABL=$ab:"; CGS_LIB_PATH=tools/variants/libcgs_clabl$ab.so CGS_LIB_ALLOW_STALE=1 python tools/ctxl_micro.py $N 20 71 2>&1 | grep "us  "
done ) | tee gpurun_out/ctxl_abl.txt
