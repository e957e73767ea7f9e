#!/bin/bash
# radix-sort tile size A/B (items per thread) on the headline step and the heavy-pair variant (GPU box)
for cfg in "$@"; do
  echo "=== $cfg"
  CGS_EXTRA_FLAGS="$cfg" python -m contextgs_amd.build > /dev/null || exit 1
  CGS_EXTRA_FLAGS="$cfg" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-eval-fps 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']; h=d['extra']['heavy_pairs']
print('headline ms', d['ms_per_step'], 'depth_sort', k['depth_sort']['avg_us'], 'tile_sort', k['tile_sort']['avg_us'], 'offsets_scan', k['offsets_scan']['avg_us'])
print('heavy ms', h['ms_per_step'], 'tile_sort', h['tile_sort_avg_us'], 'emit', h['emit_pairs_avg_us'])"
done
