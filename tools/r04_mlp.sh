# round 4 item 1: parity of the fused backward + weight-gradient kernel, then a same-box A/B against the round-3 pair
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_mlp_gpu.py tests/test_anchor_gen_gpu.py tests/test_training_parity_gpu.py tests/test_training_gpu.py -q 2>&1 | tail -8 | tee gpurun_out/r04_mlp_tests.txt
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps --steps 60"
for rep in 1 2; do
  for L in "" tools/variants/libcgs_nofuse.so; do
    CGS_LIB_PATH=$L timeout 300 python bench.py $F 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); k=j['kernels']
print('lib=${L:-product} rep=$rep', j['value'], 'views/s', j['ms_per_step'], 'ms |', ' '.join('%s %.0fus x%d' % (n, k[n]['avg_us'], k[n]['launches']//j['steps']) for n in ('mlp_fwd','mlp_bwd','mlp_wgrad')), '| mlp group', j['mlp_group_roofline']['ms_per_step'])"
  done
done | tee gpurun_out/r04_mlp_ab.txt
