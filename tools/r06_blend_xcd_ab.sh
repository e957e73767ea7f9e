# same-box A/B of the blend kernels' workgroup -> tile-order entry map: entry = workgroup id (product) against runs of R entries per
# XCD (variants rbx4 / rbx8 / rbx16: -DRB_XCD_RUN=R), headline scene and heavy-pair scene
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
FLAGS="--no-cpu-baseline --no-eval-fps --no-codec --no-raster-only --no-image-loss --steps 60"
for rep in 1 2; do for v in product "$@"; do
if [ $v = product ]; then E="X=1"; else E="CGS_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libcgs_$v.so CGS_LIB_ALLOW_STALE=1"; fi
env $E timeout 900 python bench.py $FLAGS > gpurun_out/ab.json 2> gpurun_out/bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1]); k=d["kernels"]
h=(d.get("extra") or {}).get("heavy_pairs") or {}
hk=h.get("kernels") or {}
print("$v rep $rep", d["value"], d["ms_per_step"], "kernels", d["hip_kernel_ms_per_step"], "blend_fwd", k["blend_fwd"]["avg_us"], "blend_bwd", k["blend_bwd"]["avg_us"], "heavy", d.get("value_heavy_pairs"), {n: hk[n].get("avg_us") for n in hk if "blend" in n})
PY
done; done | tee gpurun_out/r06_blend_xcd_ab.txt
