#!/bin/bash
# rocprofv3 kernel trace of a short headline bench run; summary -> gpurun_out/$1 (default prof_quick.txt)
# usage (GPU box): tools/prof_quick.sh [out-name] [extra env assignments...]
out=${1:-prof_quick.txt}; shift
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/pq && mkdir -p /tmp/pq
(cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -o pq -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps > /tmp/pq/bench.json 2> /tmp/pq/bench.err)
python $R/tools/rocprof_summary.py /tmp/pq $R/gpurun_out/$out 40 > /dev/null
tail -c 300 /tmp/pq/bench.json
head -45 $R/gpurun_out/$out | cut -c1-150
