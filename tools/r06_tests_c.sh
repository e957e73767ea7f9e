cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_raster_gpu.py tests/test_raster_edge_gpu.py tests/test_edge_cases_gpu.py -x -q 2>&1 | tail -4) > gpurun_out/r06_tc1.log 2>&1; cat gpurun_out/r06_tc1.log
(timeout 1500 python -m pytest tests/test_dist_train_gpu.py tests/test_training_gpu.py tests/test_training_parity_gpu.py tests/test_context_gpu.py -x -q -s 2>&1 | grep -E "_mask|passed|failed|Error" | tail -8) > gpurun_out/r06_tc2.log 2>&1; cat gpurun_out/r06_tc2.log
FLAGS="--no-cpu-baseline --no-heavy --no-eval-fps --no-codec --no-raster-only --no-image-loss"
timeout 600 python bench.py $FLAGS > gpurun_out/r06_bench_c.json 2> gpurun_out/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_bench_c.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["timing"]["host_ms_per_step"], d.get("hip_kernel_ms_per_step"), d["ctx_group_roofline"].get("ms_per_step"))
k=d["kernels"]
for name in ("depth_sort","offsets_scan","tile_sort","emit_pairs","ranges"):
    print(name, k.get(name))
PY
