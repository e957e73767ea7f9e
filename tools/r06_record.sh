#!/bin/bash
# Round-6 evidence run (GPU box): full GPU test suite with the allowance print-out, PMC traffic + VALU passes, the default bench line,
# rocprofv3 kernel trace of the same command, idle gaps, launch attribution.  Outputs under gpurun_out/ (copy into profiles/).
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu -x -s > gpurun_out/r06_gputests_full.log 2>&1; tail -3 gpurun_out/r06_gputests_full.log
grep -E "^\[allowance\]|^\[trajectory\]|^\[c1\]|outside|passed|failed" gpurun_out/r06_gputests_full.log | cut -c1-260 > gpurun_out/r06_gpu_tests_allowances.txt; wc -l gpurun_out/r06_gpu_tests_allowances.txt
bash tools/pmc_gpu.sh > /dev/null 2>&1; cp gpurun_out/pmc_hbm_summary.txt gpurun_out/r06_pmc_hbm.txt; head -c 700 gpurun_out/pmc_traffic.json
bash tools/pmc_valu_gen.sh 2>/dev/null | tail -1 | cut -c1-400
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json; cp gpurun_out/pmc_valu.json profiles/pmc_valu.json
timeout -k 5 1800 python bench.py > gpurun_out/r06_bench_1m.json 2> gpurun_out/r06_bench_1m.err; tail -c 800 gpurun_out/r06_bench_1m.json
bash tools/r06_prof.sh > /dev/null 2>&1; head -30 gpurun_out/r06_rocprof_bench_1m.txt | cut -c1-140; head -12 gpurun_out/r06_gpu_idle_gaps.txt
timeout 600 python tools/launch_attrib.py > /dev/null 2>&1; cp gpurun_out/launch_attrib.txt gpurun_out/r06_launch_attribution.txt; head -3 gpurun_out/r06_launch_attribution.txt
