cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
for tag in base ebf2 ebf8; do
  if [ $tag = base ]; then unset CGS_LIB_PATH; else export CGS_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libcgs_$tag.so; fi
  rm -rf /tmp/prof_e
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o e -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-heavy --no-eval-fps --no-codec --no-raster-only --no-image-loss > /dev/null 2>&1)
  python tools/rocprof_summary.py /tmp/prof_e /tmp/prof_e.txt 120 > /dev/null; echo "== $tag"; grep -E "eb_bits" /tmp/prof_e.txt | cut -c1-70
done
