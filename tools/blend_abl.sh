#!/bin/bash
# Timing ablations of blend_bwd_rows_kernel on one box (do the gathered record loads or the flush's global
# atomics bound it?).  Variants: tools/variant_lib.sh rbabl<N> raster_blend_rows.hip -DRB_ABL=<N>  (wrong results by construction):
#   1 plain LDS stores instead of LDS float atomics, 2 no flush (no global atomics), 7 records staged from a copy in LIST ORDER
#   (contiguous 48-byte loads instead of gathered ones; the copy is a kernel of its own).
# Headline scene and heavy-pair variant; per-kernel averages from rocprofv3 --kernel-trace --stats (the copy kernel of variant 7
# would count into the library's own blend_bwd scope).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-eval-fps --steps 20 --warmup 5"
for scene in headline heavy; do
 for v in ${VARIANTS:-product rbabl1 rbabl2 rbabl7}; do
  if [ $v = product ]; then E="X=1"; else E="CGS_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libcgs_$v.so CGS_LIB_ALLOW_STALE=1"; fi
  if [ $scene = heavy ]; then cmd="python $GRAFT_REPO_ROOT/tools/heavy_steps.py --steps 16"; else cmd="python $GRAFT_REPO_ROOT/bench.py $F --no-heavy"; fi
  rm -rf /tmp/ba_$v; mkdir -p /tmp/ba_$v
  (cd /tmp && env $E timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ba_$v -o p -- $cmd > /tmp/ba_$v/log.txt 2>&1)
  python tools/rocprof_summary.py /tmp/ba_$v /tmp/ba_$v/sum.txt 60 > /dev/null
  echo "== $scene $v"; grep -E "blend_bwd_rows_kernel|rb_records_in_list_order" /tmp/ba_$v/sum.txt | cut -c1-110
 done
done | tee gpurun_out/blend_abl.txt
