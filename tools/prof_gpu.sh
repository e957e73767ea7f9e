# Per-kernel rocprofv3 summary of the headline steps of bench.py (the variants that run other workloads are switched off so
# that the per-kernel averages are those of the headline configuration), next to the bench line of the same command.
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
FLAGS="--no-cpu-baseline --no-heavy --no-eval-fps --no-codec --no-raster-only --no-image-loss"
timeout 600 python bench.py $FLAGS > gpurun_out/bench_profiled_cmd.json 2> gpurun_out/bench_profiled_cmd.err
rm -rf /tmp/prof_full
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o full -- python bench.py $FLAGS > gpurun_out/prof_full.log 2>&1
python tools/rocprof_summary.py /tmp/prof_full gpurun_out/rocprof_bench_1m_summary.txt | head -40 | cut -c1-180
