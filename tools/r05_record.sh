#!/bin/bash
# Round-5 evidence run (GPU box): full GPU test suite, the default bench line, rocprofv3 kernel trace of the same command,
# PMC traffic passes, idle gaps.  Outputs under gpurun_out/ (copy the summaries into profiles/).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu -x > gpurun_out/r05_gputests.log 2>&1; tail -3 gpurun_out/r05_gputests.log
bash tools/pmc_gpu.sh > /dev/null 2>&1; cp gpurun_out/pmc_hbm_summary.txt gpurun_out/r05_pmc_hbm.txt; cp gpurun_out/pmc_traffic.json gpurun_out/r05_pmc_traffic.json; head -c 900 gpurun_out/pmc_traffic.json; cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
timeout -k 5 1500 python bench.py > gpurun_out/r05_bench_1m.json 2> gpurun_out/r05_bench_1m.err; tail -c 600 gpurun_out/r05_bench_1m.json
bash tools/r05_prof.sh > /dev/null 2>&1; head -24 gpurun_out/r05_rocprof_bench_1m.txt | cut -c1-140
timeout -k 5 600 python tools/idle_gaps.py > /dev/null 2>&1; cp gpurun_out/idle_gaps.txt gpurun_out/r05_gpu_idle_gaps.txt; head -12 gpurun_out/r05_gpu_idle_gaps.txt
