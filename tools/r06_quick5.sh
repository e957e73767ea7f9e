# raster / binning exactness, then the quick headline bench line twice + the heavy-pair leg
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_raster_gpu.py tests/test_raster_edge_gpu.py tests/test_edge_cases_gpu.py tests/test_fused_view_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | tail -4) > gpurun_out/r06_tq.log 2>&1; cat gpurun_out/r06_tq.log
FLAGS="--no-cpu-baseline --no-eval-fps --no-codec --no-raster-only --no-image-loss"
for rep in 1 2; do
timeout 900 python bench.py $FLAGS > gpurun_out/r06_bench_q.json 2> gpurun_out/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_bench_q.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "host", d["timing"]["host_ms_per_step"], "kernels", d.get("hip_kernel_ms_per_step"), "ctx", d["ctx_group_roofline"].get("ms_per_step"), "heavy", d.get("value_heavy_pairs"))
k=d["kernels"]
for name in ("depth_sort","offsets_scan","emit_pairs","tile_sort","ranges"):
    print(name, k.get(name, {}).get("avg_us"))
PY
done
