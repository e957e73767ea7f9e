# needs the experiment build of the library: CGS_EXTRA_FLAGS="-DCGS_EXPERIMENTS" python -m contextgs_amd.build (and the
# same CGS_EXTRA_FLAGS exported for the runs, it is part of the build stamp); rebuild without it afterwards
export CGS_EXTRA_FLAGS="-DCGS_EXPERIMENTS"
python -m contextgs_amd.build > /dev/null || exit 1
# blend_bwd_rows_kernel under the timing ablations CGS_ROWS_ABL=0..4 (same box, back to back); wrong gradients for != 0
cd $GRAFT_REPO_ROOT
for a in 0 1 2 4 5 6 7 0; do
  CGS_ROWS_ABL=$a timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-codec --no-heavy --no-eval-fps --no-image-loss --no-raster-only 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ABL=$a', 'blend_bwd us', d['kernels']['blend_bwd']['avg_us'], 'blend_fwd us', d['kernels']['blend_fwd']['avg_us'], 'step ms', d['ms_per_step'])"
done
