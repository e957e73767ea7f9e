# wall-clock milestones of conduct_encoding / conduct_decoding on the bench scene (CGS_CODEC_TRACE=${CGS_CODEC_TRACE:-1})
cd $GRAFT_REPO_ROOT
CGS_CODEC_TRACE=${CGS_CODEC_TRACE:-1} timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-heavy --no-eval-fps --no-raster-only --no-image-loss 2>&1 >/dev/null | grep -E "^\[(encode|decode)|ing time" | tail -40
