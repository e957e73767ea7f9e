# context / training / trajectory / dist tests, then A/B of the early level launches (CGS_EARLY_LEVELS) on the quick headline bench + idle gaps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_context_gpu.py tests/test_training_parity_gpu.py tests/test_rate_sub_gpu.py tests/test_ctx_level_gpu.py tests/test_training_gpu.py tests/test_ctx_ops_gpu.py tests/test_edge_cases_gpu.py tests/test_dist_train_gpu.py tests/test_trajectory_gpu.py -x -q 2>&1 | tail -5) > gpurun_out/r06_tq.log 2>&1; cat gpurun_out/r06_tq.log
FLAGS="--no-cpu-baseline --no-heavy --no-eval-fps --no-codec --no-raster-only --no-image-loss"
for rep in 1 2; do for e in 11; do
CGS_EARLY_LEVELS=${e:0:1} CGS_EARLY_MLP3=${e:1:1} timeout 600 python bench.py $FLAGS > gpurun_out/r06_bench_q.json 2> gpurun_out/bench.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r06_bench_q.json").read().strip().splitlines()[-1])
print("early(levels,mlp3)=$e value", d["value"], "ms", d["ms_per_step"], "host", d["timing"]["host_ms_per_step"], "kernels", d.get("hip_kernel_ms_per_step"), "ctx", d["ctx_group_roofline"].get("ms_per_step"))
PY
done; done
timeout -k 5 600 python tools/idle_gaps.py > /dev/null 2>&1; head -24 gpurun_out/idle_gaps.txt | cut -c1-200
