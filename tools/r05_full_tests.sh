#!/bin/bash
# the whole GPU suite as shipped, then the rasterizer / training / trajectory files again with the two-level binning forced wherever
# the grid allows it (CGS_BIN_MODE=2) -> gpurun_out/r05_full_tests.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( timeout -k 10 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -6
  echo "== CGS_BIN_MODE=2"
  CGS_BIN_MODE=2 timeout -k 10 1200 python -m pytest tests/test_raster_gpu.py tests/test_raster_edge_gpu.py tests/test_training_gpu.py tests/test_trajectory_gpu.py tests/test_context_gpu.py tests/test_edge_cases_gpu.py -q -x 2>&1 | tail -4 ) | tee gpurun_out/r05_full_tests.txt
