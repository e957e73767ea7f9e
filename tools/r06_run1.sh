cd /root/repo
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_rate_sub_gpu.py -x -q -s 2>&1 | tail -60) > gpurun_out/r06_t1.log
(timeout 300 python tools/rate_sub_micro.py 2>&1 | tail -20) > gpurun_out/r06_micro1.log
(timeout 900 python -m pytest tests/test_ctx_level_gpu.py tests/test_training_parity_gpu.py tests/test_context_gpu.py -x -q 2>&1 | tail -15) > gpurun_out/r06_t2.log
cat gpurun_out/r06_t1.log gpurun_out/r06_micro1.log gpurun_out/r06_t2.log
