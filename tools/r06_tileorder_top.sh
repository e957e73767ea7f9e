# blend kernels' tile order: how many list-length classes matter?  TO_TOP = c: every list of >= 2^(c-1) entries in one class
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FLAGS="--no-cpu-baseline --no-eval-fps --no-codec --no-raster-only --no-image-loss --no-heavy"
for rep in 1 2; do for v in product top11 top10 top9; do
if [ $v = product ]; then E="X=1"; else E="CGS_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libcgs_$v.so CGS_LIB_ALLOW_STALE=1"; fi
env $E timeout 900 python bench.py $FLAGS > gpurun_out/ab.json 2> gpurun_out/bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1]); k=d["kernels"]
print("$v rep $rep", d["value"], d["ms_per_step"], "kernels", d["hip_kernel_ms_per_step"], "blend_fwd", k["blend_fwd"]["avg_us"], "blend_bwd", k["blend_bwd"]["avg_us"])
PY
done; done | tee gpurun_out/r06_tile_order_top.txt
