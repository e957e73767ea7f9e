#!/bin/bash
# MLP kernel tilings (rows per wave tile RT x waves per workgroup) A/B on the headline step (GPU box).
# usage: tools/mlp_tiling.sh "<flags>" ...   e.g. "-DM3_FWD_RT=2 -DM3_FWD_WAVES=8" "-DM2_RT=2 -DM2_WAVES=8" "-DM3_BWD_RT=1 -DM3_BWD_WAVES=16"
for cfg in "$@"; do
  echo "=== $cfg"
  CGS_EXTRA_FLAGS="$cfg" python -m contextgs_amd.build > /dev/null || exit 1
  CGS_EXTRA_FLAGS="$cfg" tools/prof_quick.sh pq_tiling.txt CGS_EXTRA_FLAGS="$cfg" | grep -E "mlp3_|mlp2_|wgrad_multi_kernel|ms_per_step" | cut -c1-120
  grep -o '"ms_per_step": [0-9.]*' /tmp/pq/bench.json | head -1
done
