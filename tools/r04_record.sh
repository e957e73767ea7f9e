#!/bin/bash
# Round-4 evidence run (GPU box): full GPU test suite, the default bench line, rocprofv3 kernel trace of the same command,
# PMC traffic passes.  Outputs under gpurun_out/ (copy the summaries into profiles/).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu -x > gpurun_out/r04_gputests.log 2>&1; tail -3 gpurun_out/r04_gputests.log
python bench.py > gpurun_out/r04_bench_1m.json 2> gpurun_out/r04_bench_1m.err; tail -c 600 gpurun_out/r04_bench_1m.json
export TMPDIR=/tmp
rm -rf /tmp/prof_full && mkdir -p /tmp/prof_full
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-codec --no-heavy --no-eval-fps > $GRAFT_REPO_ROOT/gpurun_out/r04_bench_1m_profiled_cmd.json 2> /dev/null)
python tools/rocprof_summary.py /tmp/prof_full gpurun_out/r04_rocprof_bench_1m.txt 60 > /dev/null; head -30 gpurun_out/r04_rocprof_bench_1m.txt | cut -c1-140
bash tools/pmc_gpu.sh > /dev/null 2>&1; cp gpurun_out/pmc_hbm_summary.txt gpurun_out/r04_pmc_hbm.txt; cp gpurun_out/pmc_traffic.json gpurun_out/r04_pmc_traffic.json; cat gpurun_out/pmc_traffic.json
python tools/idle_gaps.py > /dev/null 2>&1; cp gpurun_out/idle_gaps.txt gpurun_out/r04_gpu_idle_gaps.txt; head -12 gpurun_out/r04_gpu_idle_gaps.txt
