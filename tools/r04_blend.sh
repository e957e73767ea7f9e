# round 4 item 3: parity of the blend backward without LDS float atomics, then same-box A/B against round 3's kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
true
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-eval-fps --steps 60"
for rep in 1 2 3; do
  for tag in product rawacc; do
    if [ $tag = product ]; then L=""; else L=tools/variants/libcgs_$tag.so; fi
    CGS_LIB_PATH=$L timeout 300 python bench.py $F 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); k=j['kernels']; h=j['extra'].get('heavy_pairs',{})
print('lib=$tag rep=$rep', j['value'], 'views/s', j['ms_per_step'], 'ms |', ' '.join('%s %.0fus' % (n, k[n]['avg_us']) for n in ('blend_bwd','blend_fwd','preprocess_bwd')), '| hip kernels', j['hip_kernel_ms_per_step'], '| heavy', h.get('value'), h.get('blend_bwd_avg_us'))"
  done
done | tee gpurun_out/r04_blend_ab.txt
