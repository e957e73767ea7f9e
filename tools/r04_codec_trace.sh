cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CGS_CODEC_TRACE=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-heavy --no-eval-fps --no-raster-only --no-image-loss > gpurun_out/r04_codec_bench.json 2> gpurun_out/r04_codec_trace.txt
grep -E "^\[(decode|context_rows)" gpurun_out/r04_codec_trace.txt | tail -36
