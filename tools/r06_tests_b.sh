cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_dist_train_gpu.py tests/test_dist_codec_gpu.py -x -q -s 2>&1 | grep -E "_mask|passed|failed|Error|rank" | tail -15) > gpurun_out/r06_tb.log 2>&1
cat gpurun_out/r06_tb.log
(timeout 1200 python bench.py --no-cpu-baseline --no-heavy --no-raster-only --no-image-loss > gpurun_out/r06_bench_codec.json 2> gpurun_out/r06_bench_codec.err; tail -3 gpurun_out/r06_bench_codec.err)
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_bench_codec.json").read().strip().splitlines()[-1])
c=d["codec"]
print(d["value"], d["ms_per_step"])
for k in ("encode_Manchors_per_s","decode_Manchors_per_s","decoded_feat_scaling_offsets_hyper_bit_exact_vs_encoder_quantised","max_over_median"): print(k, c.get(k), c["container_v2"].get(k))
print(json.dumps(c["c3_500k"]))
PY
