#!/bin/bash
# tools/r06_kernel_ab.sh <kernel name regex> <lib or "-" for the product build> ...: per-kernel average durations of the headline
# bench (30 steps) under rocprofv3 --kernel-trace --stats for each library variant, same box, two alternating rounds.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
pat=$1; shift
FLAGS="--no-cpu-baseline --no-codec --no-heavy --no-eval-fps --no-raster-only --no-image-loss --steps 30 --warmup 5"
for rep in 1 2; do for lib in "$@"; do
  rm -rf /tmp/prof_ab && mkdir -p /tmp/prof_ab
  if [ "$lib" = "-" ]; then unset CGS_LIB_PATH CGS_LIB_ALLOW_STALE; else export CGS_LIB_PATH=$GRAFT_REPO_ROOT/$lib CGS_LIB_ALLOW_STALE=1; fi
  (cd /tmp && timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o p -- python $GRAFT_REPO_ROOT/bench.py $FLAGS > /tmp/prof_ab/bench.json 2> /dev/null)
  python tools/rocprof_summary.py /tmp/prof_ab /tmp/prof_ab/sum.txt 200 > /dev/null
  echo "== rep $rep lib $lib: $(python -c "import json;d=json.loads(open('/tmp/prof_ab/bench.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'])")"
  grep -E "$pat" /tmp/prof_ab/sum.txt | cut -c1-110
done; done
