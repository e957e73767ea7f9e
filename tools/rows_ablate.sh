#!/bin/bash
# Timing ablations of blend_bwd_rows_kernel (wrong gradients by construction).  Round 3's kernel (per-block entry lists):
# -DRB_ABL=1 plain LDS stores instead of float atomics, =2 no flush to global memory, =3 no row reduction and no accumulation,
# =4 walk + evaluation only.  Build the four variant libraries in the authoring container, then run this on the GPU box:
#   for k in 1 2 3 4; do tools/variant_lib.sh rbabl$k raster_blend_rows.hip -DRB_ABL=$k; done
# (Round 2's kernel and its CGS_ROWS_ABL=1..7 switch are in the history: profiles/r02_blend_bwd_experiments.txt.)
cd $GRAFT_REPO_ROOT
bash tools/variants_prof.sh "blend_bwd" rbabl1 rbabl2 rbabl3 rbabl4
