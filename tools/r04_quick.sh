# quick GPU check: the MLP / training / context tests, then two short bench lines (kernel groups)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_mlp_gpu.py tests/test_anchor_gen_gpu.py tests/test_training_parity_gpu.py tests/test_training_gpu.py tests/test_ctx_ops_gpu.py tests/test_api_edge_gpu.py tests/test_edge_cases_gpu.py tests/test_dist_train_gpu.py -q 2>&1 | tail -6 | tee gpurun_out/r04_quick_tests.txt
F="--no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps --steps 60"
for rep in 1 2; do
  timeout 300 python bench.py $F 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); k=j['kernels']
print('rep=$rep', j['value'], 'views/s', j['ms_per_step'], 'ms |', ' '.join('%s %.0fus x%d' % (n, k[n]['avg_us'], k[n]['launches']//j['steps']) for n in ('mlp_fwd','mlp_bwd','mlp_wgrad','ctx_fwd','ctx_bwd')), '| mlp group', j['mlp_group_roofline']['ms_per_step'], '| hip kernels', j['hip_kernel_ms_per_step'])"
done | tee gpurun_out/r04_quick_bench.txt
