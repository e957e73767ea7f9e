cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_rate_sub_gpu.py -x -q 2>&1 | tail -3) > gpurun_out/r06_t1.log
(timeout 300 python tools/rate_sub_micro.py 2>&1 | grep fused) > gpurun_out/r06_micro3.log
FLAGS="--no-cpu-baseline --no-heavy --no-eval-fps --no-codec --no-raster-only --no-image-loss"
for rep in 1 2; do
CGS_RATE_FUSED=0 timeout 600 python bench.py $FLAGS > gpurun_out/r06_bench_ratefused0_$rep.json 2> gpurun_out/bench.err
timeout 600 python bench.py $FLAGS > gpurun_out/r06_bench_ratefused1_$rep.json 2>> gpurun_out/bench.err
done
cat gpurun_out/r06_t1.log gpurun_out/r06_micro3.log
python - <<'PY'
import json
for n in ("ratefused0_1","ratefused1_1","ratefused0_2","ratefused1_2"):
    try:
        d=json.loads(open(f"gpurun_out/r06_bench_{n}.json").read().strip().splitlines()[-1])
        t=d.get("timing",{})
        print(n, d["value"], d["ms_per_step"], {k:t.get(k) for k in ("hip_kernel_ms_per_step","host_ms_per_step","launches_per_step")}, d.get("ctx_group_roofline",{}).get("ms_per_step"), d.get("ctx_group_roofline",{}).get("launches"))
    except Exception as e:
        print(n, "ERR", e)
PY
tail -5 gpurun_out/bench.err
