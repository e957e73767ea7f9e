#!/bin/bash
# rocprofv3 kernel table of the binning kernels, headline scene and heavy-pair variant, radix passes (mode 1) against the
# two-level binning (mode 2) -> gpurun_out/bin_prof_<scene>_m<mode>.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "" "--voxel 0.01"; do
 for m in 1 2; do
  tag=$( [ -z "$v" ] && echo headline || echo heavy )_m$m
  rm -rf /tmp/bp_$tag; mkdir -p /tmp/bp_$tag
  (cd /tmp && timeout -k 5 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp_$tag -o p -- python $GRAFT_REPO_ROOT/tools/bin_prof.py $v --mode $m > $GRAFT_REPO_ROOT/gpurun_out/bin_prof_$tag.log 2>&1)
  python tools/rocprof_summary.py /tmp/bp_$tag gpurun_out/bin_prof_$tag.txt 40 > /dev/null
  echo "== $tag"; tail -1 gpurun_out/bin_prof_$tag.log
  grep -E "tb_|bk_|radix|scan_|gather_rects|block_first|ranges|emit|iota|blend_fwd|preprocess" gpurun_out/bin_prof_$tag.txt | cut -c1-150
 done
done
for v in "" "--voxel 0.01"; do CGS_BK_TRACE=1 timeout 120 python tools/bin_prof.py $v --mode 2 --iters 1 2>&1 | grep "\[bk\] P"; done
