# per-kernel blend times of the heavy-pair scene under both workgroup -> tile maps (rocprofv3 kernel trace)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in product rbraster; do
  if [ $v = product ]; then E="X=1"; else E="CGS_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libcgs_$v.so CGS_LIB_ALLOW_STALE=1"; fi
  rm -rf /tmp/to_$v; mkdir -p /tmp/to_$v
  (cd /tmp && env $E timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/to_$v -o p -- python $GRAFT_REPO_ROOT/tools/heavy_steps.py --steps 16 > /tmp/to_$v/log.txt 2>&1) || true
  python tools/rocprof_summary.py /tmp/to_$v /tmp/to_$v/sum.txt 60 > /dev/null
  echo "== heavy $v"; grep -E "blend_|tile_order|bk_fill|bk_count" /tmp/to_$v/sum.txt | cut -c1-100; tail -2 /tmp/to_$v/log.txt
done | tee gpurun_out/r06_tile_order_heavy.txt
