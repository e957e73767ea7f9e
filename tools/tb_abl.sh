#!/bin/bash
# timing ablations of the tile-binning scatter kernels (wrong results): TB_ABL=1 no digit match (identity placement),
# =2 no global stores.  Variant libraries are built in the authoring container: tools/variant_lib.sh tbabl1 tile_bin.hip -DTB_ABL=1 ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for abl in 0 1 2; do
  if [ $abl = 0 ]; then unset CGS_LIB_PATH; else export CGS_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libcgs_tbabl$abl.so; fi
  rm -rf /tmp/tba; mkdir -p /tmp/tba
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tba -o p -- python $GRAFT_REPO_ROOT/tools/bin_prof.py --voxel 0.01 --iters 6 > /dev/null 2>&1)
  python tools/rocprof_summary.py /tmp/tba /tmp/tba.txt 40 > /dev/null
  echo "== TB_ABL=$abl"; grep -E "tb_" /tmp/tba.txt | cut -c1-110
done
