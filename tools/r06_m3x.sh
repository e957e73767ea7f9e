cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_mlp_gpu.py tests/test_anchor_gen_gpu.py tests/test_training_gpu.py tests/test_fused_view_gpu.py tests/test_training_parity_gpu.py -x -q 2>&1 | tail -3) > gpurun_out/r06_tm.log 2>&1; cat gpurun_out/r06_tm.log
FLAGS="--no-cpu-baseline --no-heavy --no-eval-fps --no-codec --no-raster-only --no-image-loss"
for rep in 1 2; do
for keep in 1 0; do
CGS_M3_KEEP_X=$keep timeout 600 python bench.py $FLAGS > gpurun_out/r06_bench_keepx$keep.json 2> gpurun_out/bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_keepx$keep.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("keep_x=$keep value", d["value"], "ms", d["ms_per_step"], "kernels", d.get("hip_kernel_ms_per_step"), "mlp_fwd", k.get("mlp_fwd",{}).get("avg_us"), "mlp_bwd", k.get("mlp_bwd",{}).get("avg_us"))
PY
done
done
