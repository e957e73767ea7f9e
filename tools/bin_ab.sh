#!/bin/bash
# Tile binning A/B on one box: the rasterizer tests, then the headline step and the heavy-pair variant with the radix passes
# (CGS_BIN_MODE=1), the two-level binning wherever the grid allows it (2) and the per-view choice (0 = what ships).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 5 900 python -m pytest tests/test_raster_gpu.py tests/test_raster_edge_gpu.py tests/test_api_edge_gpu.py -x -q 2>&1 | tail -4 | tee gpurun_out/bin_ab_tests.txt
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-eval-fps --steps 60"
for rep in 1 2; do
 for m in 1 2 0; do
  CGS_BIN_MODE=$m timeout -k 5 400 python bench.py $F 2>gpurun_out/bin_ab_err_$m.txt | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); k=j['kernels']; h=j.get('extra',{}).get('heavy_pairs') or {}
print('mode=$m rep=$rep headline', j['value'], 'views/s', j['ms_per_step'], 'ms |', ' '.join('%s %.0fus' % (n, k[n]['avg_us']) for n in ('emit_pairs','tile_sort','ranges') if n in k),
      '| heavy', h.get('value'), 'views/s', h.get('ms_per_step'), 'ms emit', h.get('emit_pairs_avg_us'), 'sort', h.get('tile_sort_avg_us'), 'blend', h.get('blend_fwd_avg_us'), h.get('blend_bwd_avg_us'))"
 done
done | tee gpurun_out/bin_ab.txt
