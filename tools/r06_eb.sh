cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_eb_gpu.py tests/test_context_gpu.py tests/test_training_parity_gpu.py tests/test_trajectory_gpu.py tests/test_entropy_api_gpu.py tests/test_codec_gpu.py -x -q 2>&1 | tail -3) > gpurun_out/r06_te.log 2>&1; cat gpurun_out/r06_te.log
rm -rf /tmp/prof_e; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o e -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-heavy --no-eval-fps --no-codec --no-raster-only --no-image-loss > $GRAFT_REPO_ROOT/gpurun_out/r06_bench_e.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py /tmp/prof_e gpurun_out/r06_prof_e.txt 90 > /dev/null; grep -E "eb_bits|ctx_choose|hyper_noise" gpurun_out/r06_prof_e.txt | cut -c1-100
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_bench_e.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
PY
