# same-box A/B of the anchor-MLP backward variants (round 4 item 1): tools/r04_mlp_ab.sh <reps> <lib tags ...> ("product" = in-tree)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
reps=$1; shift
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps --steps 60"
for rep in $(seq 1 $reps); do
  for tag in "$@"; do
    if [ $tag = product ]; then L=""; else L=tools/variants/libcgs_$tag.so; fi
    CGS_LIB_PATH=$L timeout 300 python bench.py $F 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); k=j['kernels']
print('lib=$tag rep=$rep', j['value'], 'views/s', j['ms_per_step'], 'ms | profiled pass', j['ms_per_step_profiled_pass'], '|', ' '.join('%s %.0fus x%d' % (n, k[n]['avg_us'], k[n]['launches']//j['steps']) for n in ('mlp_fwd','mlp_bwd','mlp_wgrad')), '| mlp group', j['mlp_group_roofline']['ms_per_step'], '| hip kernels', j['hip_kernel_ms_per_step'])"
  done
done | tee gpurun_out/r04_mlp_ab.txt
