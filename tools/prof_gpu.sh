set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --steps 5 --warmup 2 > gpurun_out/bench_1m.json 2> gpurun_out/bench_1m.err; tail -c 2500 gpurun_out/bench_1m.json
rm -rf /tmp/prof_full
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o full -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_full.log 2>&1
python tools/rocprof_summary.py /tmp/prof_full gpurun_out/rocprof_bench_1m_summary.txt | head -40 | cut -c1-180
