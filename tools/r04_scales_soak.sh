cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/scales.sh 2>&1 | tee gpurun_out/r04_scales.txt
timeout 900 python tools/soak.py 1000 2>&1 | tail -25 | tee gpurun_out/r04_soak.txt
