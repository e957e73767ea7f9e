# round 4, item 2 / 4: container goldens, codec tests, and the v1 / v2 codec timelines of the bench scene
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/golden
timeout 600 python tools/make_container_golden.py gpurun_out/golden > gpurun_out/golden/log.txt 2>&1
tail -3 gpurun_out/golden/log.txt
cp gpurun_out/golden/container_n*.json tests/golden/ 2>/dev/null
timeout 1500 python -m pytest tests/test_codec_gpu.py tests/test_codec.py tests/test_configs_gpu.py -x -q 2>&1 | tail -15 | tee gpurun_out/r04_codec_tests.txt
CGS_CODEC_TRACE=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-heavy --no-eval-fps --no-raster-only --no-image-loss > gpurun_out/r04_codec_bench.json 2> gpurun_out/r04_codec_trace.txt
python - <<'PY'
import json
r = json.load(open("gpurun_out/r04_codec_bench.json"))
print(json.dumps(r["codec"], indent=1))
PY
grep -E "^\[(encode|decode)|ing time|codec time" gpurun_out/r04_codec_trace.txt | tail -120
