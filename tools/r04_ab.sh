# generic same-box A/B of library variants on the headline step: tools/r04_ab.sh <reps> <tag ...> ("product" = in-tree library)
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
print('lib=$tag rep=$rep', j['value'], 'views/s', j['ms_per_step'], 'ms |', ' '.join('%s %.0fus x%d' % (n, k[n]['avg_us'], k[n]['launches']//j['steps']) for n in ('mlp_fwd','mlp_bwd','mlp_wgrad','ctx_fwd','ctx_bwd','rate_fwd','rate_bwd')), '| hip kernels', j['hip_kernel_ms_per_step'])"
  done
done | tee gpurun_out/r04_ab.txt
