#!/bin/bash
# rocprofv3 kernel table of the binning kernels, headline scene and heavy-pair variant -> gpurun_out/bin_prof_*.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "" "--voxel 0.01"; do
  tag=$( [ -z "$v" ] && echo headline || echo heavy )
  rm -rf /tmp/bp_$tag; mkdir -p /tmp/bp_$tag
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp_$tag -o p -- python $GRAFT_REPO_ROOT/tools/bin_prof.py $v > $GRAFT_REPO_ROOT/gpurun_out/bin_prof_$tag.log 2>&1)
  python tools/rocprof_summary.py /tmp/bp_$tag gpurun_out/bin_prof_$tag.txt 40 > /dev/null
  grep -E "tb_|radix|scan_|gather_rects|block_first|ranges|emit|iota|blend_fwd|preprocess" gpurun_out/bin_prof_$tag.txt | cut -c1-150
done
