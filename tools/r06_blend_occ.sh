#!/bin/bash
# Round 6, VERDICT r5 item 4 (blend backward without the LDS float-atomic unit): what a per-(block, entry) slot pool may cost in
# occupancy.  The slot pool needs ~30-36 KB of LDS per workgroup on top of the kernel's 22.6 KB (7 workgroups per CU today, 3 or
# 2 with the pool).  This script measures the SHIPPED kernel at those occupancies: the same code launched with 8 / 18 / 32 KB of
# unused dynamic LDS (no source branch: the launch line of a scratch copy of raster_blend_rows.hip is rewritten by sed).
#   authoring container:  bash tools/r06_blend_occ.sh build      GPU box:  bash tools/r06_blend_occ.sh run
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  python -c "import __graft_entry__ as g; g.build()" > /dev/null
  mkdir -p tools/variants/src
  for kb in 8 18 32; do
    src=tools/variants/src/raster_blend_rows_dyn$kb.hip
    # only the BACKWARD launch (the second hipLaunchKernelGGL of the file) gets the dynamic LDS
    awk -v kb=$kb 'BEGIN{n=0} /hipLaunchKernelGGL\(blend_bwd_rows_kernel/ {sub(/dim3\(RB_THREADS\), 0, stream/, "dim3(RB_THREADS), " kb*1024 ", stream")} {print}' contextgs_amd/csrc/raster_blend_rows.hip > $src
    grep -c "dim3(RB_THREADS), $((kb*1024)), stream" $src
    hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude -Icontextgs_amd/csrc -c $src -o tools/variants/rbdyn$kb.o
    others=$(ls contextgs_amd/csrc/build/*.o | grep -v "/raster_blend_rows.hip.o")
    hipcc -shared -fPIC --offload-arch=gfx950 $others tools/variants/rbdyn$kb.o -o tools/variants/libcgs_rbdyn$kb.so
  done
  exit 0
fi
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-eval-fps --steps 20 --warmup 5"
for scene in headline heavy; do
 for v in product rbdyn8 rbdyn18 rbdyn32; do
  if [ $v = product ]; then E="X=1"; else E="CGS_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libcgs_$v.so CGS_LIB_ALLOW_STALE=1"; fi
  if [ $scene = heavy ]; then cmd="python $GRAFT_REPO_ROOT/tools/heavy_steps.py --steps 16"; else cmd="python $GRAFT_REPO_ROOT/bench.py $F --no-heavy"; fi
  rm -rf /tmp/bo_$v; mkdir -p /tmp/bo_$v
  (cd /tmp && env $E timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bo_$v -o p -- $cmd > /tmp/bo_$v/log.txt 2>&1) || true
  python tools/rocprof_summary.py /tmp/bo_$v /tmp/bo_$v/sum.txt 60 > /dev/null
  echo "== $scene $v"; grep -E "blend_bwd_rows_kernel" /tmp/bo_$v/sum.txt | cut -c1-110
 done
done | tee gpurun_out/r06_blend_occupancy.txt
