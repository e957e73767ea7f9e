# hyper latents' gradient rows written by the level kernels (CGS_HYPER_DIRECT): tests, then same-box A/B on the quick headline bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_ctx_level_gpu.py tests/test_training_gpu.py tests/test_training_parity_gpu.py tests/test_context_gpu.py tests/test_trajectory_gpu.py tests/test_dist_train_gpu.py tests/test_eb_gpu.py tests/test_rate_sub_gpu.py -x -q 2>&1 | tail -5)
FLAGS="--no-cpu-baseline --no-heavy --no-eval-fps --no-codec --no-raster-only --no-image-loss --steps 60"
for rep in 1 2; do for e in 1 0; do
CGS_HYPER_DIRECT=$e timeout 600 python bench.py $FLAGS > gpurun_out/hd.json 2> gpurun_out/bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/hd.json").read().strip().splitlines()[-1])
k=d["kernels"]
print("hyper_direct=$e value", d["value"], "ms", d["ms_per_step"], "kernels", d.get("hip_kernel_ms_per_step"), "ctx", d["ctx_group_roofline"].get("ms_per_step"), "launches", d.get("launches_per_step"), {n: k[n]["avg_us"] for n in k if "ctx_bwd" in n or "level" in n})
PY
done; done
