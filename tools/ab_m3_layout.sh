cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for tag in product m3packed product m3packed; do
  if [ $tag = product ]; then unset CGS_LIB_PATH; else export CGS_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libcgs_$tag.so; fi
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('$tag', d['value'], d['ms_per_step'], d['hip_kernel_ms_per_step'], 'mlp f/b/w', k['mlp_fwd']['total_ms'], k['mlp_bwd']['total_ms'], k['mlp_wgrad']['total_ms'], 'expand', k['expand_fwd']['avg_us'], k['expand_bwd']['avg_us'], 'blend', k['blend_fwd']['avg_us'], k['blend_bwd']['avg_us'])"
done
