cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_trajectory_gpu.py tests/test_configs_gpu.py::test_c1_10k_anchors_256x256_vs_oracle -x -q -s 2>&1 | grep -E "^\[trajectory\] (final|worst)|^\[c1\]|passed|failed|Error|assert" | head -30) > gpurun_out/r06_ta.log
cat gpurun_out/r06_ta.log
(timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -8) > gpurun_out/r06_full_a.log
cat gpurun_out/r06_full_a.log
