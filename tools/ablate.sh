# needs the experiment build of the library: CGS_EXTRA_FLAGS="-DCGS_EXPERIMENTS" python -m contextgs_amd.build (and the
# same CGS_EXTRA_FLAGS exported for the runs, it is part of the build stamp); rebuild without it afterwards
export CGS_EXTRA_FLAGS="-DCGS_EXPERIMENTS"
python -m contextgs_amd.build > /dev/null || exit 1
python -m pytest tests/test_raster_gpu.py -x -q 2>&1 | tail -3
for a in 0 2 3; do
  CGS_BWD_ABLATE=$a python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-raster-only --step-semantics 1000 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('ablate $a: blend_bwd', k['blend_bwd']['avg_us'], 'us  blend_fwd', k['blend_fwd']['avg_us'], ' step ms', d['ms_per_step'])"
done
