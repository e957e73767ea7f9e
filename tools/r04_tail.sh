cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_training_parity_gpu.py -q 2>&1 | tail -3
bash tools/r04_ab.sh 2 product tail3 notail
