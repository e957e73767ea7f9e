# same-box A/B of the tile binning's XCD-aware column assignment (product) against column = workgroup id (variant tbnoxcd)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FLAGS="--no-cpu-baseline --no-eval-fps --no-codec --no-raster-only --no-image-loss --no-heavy"
for rep in 1 2 3; do for v in product tbnoxcd; do
if [ $v = product ]; then E="X=1"; else E="CGS_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libcgs_$v.so CGS_LIB_ALLOW_STALE=1"; fi
env $E timeout 600 python bench.py $FLAGS > gpurun_out/ab.json 2> gpurun_out/bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("$v rep $rep", d["value"], d["ms_per_step"], "kernels", d["hip_kernel_ms_per_step"], "emit_pairs", k["emit_pairs"]["avg_us"], "tile_sort", k["tile_sort"]["avg_us"], "depth_sort", k["depth_sort"]["avg_us"])
PY
done; done | tee gpurun_out/r06_tb_xcd_ab.txt
