cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_rate_sub_gpu.py tests/test_context_gpu.py tests/test_entropy_api_gpu.py -x -q 2>&1 | tail -8) > gpurun_out/r06_t1.log
(timeout 300 python tools/rate_sub_micro.py 2>&1 | grep fused) > gpurun_out/r06_micro2.log
cat gpurun_out/r06_t1.log gpurun_out/r06_micro2.log
