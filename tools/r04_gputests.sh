cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee gpurun_out/r04_gputests_tail.txt
