cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_h
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_h -o h -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-raster-only --no-codec --no-image-loss > /dev/null 2>&1
for p in "CUDAFunctor_add<float>" "FillFunctor<float>" "direct_copy_kernel" "MulFunctor"; do python tools/dispatch_hist.py /tmp/prof_h "$p" 12 | head -9; done
