#!/bin/bash
# timing ablations of ag_bwd_kernel (GPU box): rebuilds the library with -DAG_ABL=k, profiles, restores nothing (scratch copy)
for k in "$@"; do
  CGS_EXTRA_FLAGS="-DCGS_EXPERIMENTS -DAG_ABL=$k" python -m contextgs_amd.build > /dev/null || exit 1
  CGS_EXTRA_FLAGS="-DCGS_EXPERIMENTS -DAG_ABL=$k" tools/prof_quick.sh pq_abl$k.txt CGS_EXTRA_FLAGS="-DCGS_EXPERIMENTS -DAG_ABL=$k" | grep -E "ag_|wgrad_multi_kernel|mlp3"
done
