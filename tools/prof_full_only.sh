cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/prof_f
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -o f -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-raster-only --no-codec --no-image-loss > gpurun_out/prof_f.log 2>&1
python tools/rocprof_summary.py /tmp/prof_f gpurun_out/rocprof_full_only.txt 60 > /dev/null
tail -c 400 gpurun_out/prof_f.log | head -c 300
