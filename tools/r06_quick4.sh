# sort / raster exactness + context tests, then the quick headline bench line twice
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_raster_gpu.py tests/test_knn_gpu.py tests/test_context_gpu.py tests/test_training_parity_gpu.py tests/test_training_gpu.py tests/test_trajectory_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r06_tq.log 2>&1; cat gpurun_out/r06_tq.log
FLAGS="--no-cpu-baseline --no-heavy --no-eval-fps --no-codec --no-raster-only --no-image-loss"
for rep in 1 2; do
timeout 600 python bench.py $FLAGS > gpurun_out/r06_bench_q.json 2> gpurun_out/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_bench_q.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "host", d["timing"]["host_ms_per_step"], "kernels", d.get("hip_kernel_ms_per_step"), "ctx", d["ctx_group_roofline"].get("ms_per_step"))
k=d["kernels"]
for name in ("depth_sort","offsets_scan"):
    print(name, k.get(name))
PY
done
timeout 600 python tools/launch_attrib.py > /dev/null 2>&1; head -1 gpurun_out/launch_attrib.txt
