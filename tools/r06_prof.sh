#!/bin/bash
# rocprofv3 kernel trace of the headline bench command (no codec / cpu baseline / heavy / eval legs) -> gpurun_out/r06_rocprof_bench_1m.txt, + idle gaps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/prof_full && mkdir -p /tmp/prof_full
(cd /tmp && timeout -k 5 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-codec --no-heavy --no-eval-fps --no-raster-only --no-image-loss $@ > $GRAFT_REPO_ROOT/gpurun_out/r06_bench_1m_profiled_cmd.json 2> /dev/null)
python tools/rocprof_summary.py /tmp/prof_full gpurun_out/r06_rocprof_bench_1m.txt 90 > /dev/null; head -75 gpurun_out/r06_rocprof_bench_1m.txt | cut -c1-150
timeout -k 5 600 python tools/idle_gaps.py > /dev/null 2>&1; cp gpurun_out/idle_gaps.txt gpurun_out/r06_gpu_idle_gaps.txt; head -30 gpurun_out/r06_gpu_idle_gaps.txt | cut -c1-200
