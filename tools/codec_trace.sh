# wall-clock milestones of conduct_encoding / conduct_decoding on the bench scene (CGS_CODEC_TRACE=1; =2 drains the device
# at every milestone, which attributes device time); full output in gpurun_out/codec_trace.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
CGS_CODEC_TRACE=${CGS_CODEC_TRACE:-1} timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-heavy --no-eval-fps --no-raster-only --no-image-loss 2>&1 >/dev/null | grep -E "^\[(encode|decode)|ing time" > gpurun_out/codec_trace.txt
# the second (warm) run of each direction
awk '/^encoding time/{e++} /^decoding time/{d++} {print}' gpurun_out/codec_trace.txt | tail -70
