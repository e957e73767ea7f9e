# same-box A/B of the blend kernels' workgroup -> tile map: longest lists first (product) against raster order (variant rbraster),
# headline scene and heavy-pair scene
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_raster_edge_gpu.py tests/test_fused_view_gpu.py -x -q 2>&1 | tail -2)
FLAGS="--no-cpu-baseline --no-eval-fps --no-codec --no-raster-only --no-image-loss"
for rep in 1 2; do for v in product rbraster; do
if [ $v = product ]; then E="X=1"; else E="CGS_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libcgs_$v.so CGS_LIB_ALLOW_STALE=1"; fi
env $E timeout 900 python bench.py $FLAGS > gpurun_out/ab.json 2> gpurun_out/bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1]); k=d["kernels"]
h=d.get("extra",{}).get("heavy_pairs",{}) if isinstance(d.get("extra"),dict) else {}
print("$v rep $rep", d["value"], d["ms_per_step"], "kernels", d["hip_kernel_ms_per_step"], "blend_fwd", k["blend_fwd"]["avg_us"], "blend_bwd", k["blend_bwd"]["avg_us"], "heavy", d.get("value_heavy_pairs"))
PY
done; done | tee gpurun_out/r06_tile_order_ab.txt
