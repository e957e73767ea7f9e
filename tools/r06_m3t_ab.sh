#!/bin/bash
# tools/r06_m3t_ab.sh <lib or "-"> ...: mlp3_fwd / mlp3_bwd_wg under rocprofv3 --kernel-trace --stats for CGS_M3_TILED=0 / 1 with the product
# build ("-") and for each variant library (CGS_M3_TILED=1), same box, two alternating rounds.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
FLAGS="--no-cpu-baseline --no-codec --no-heavy --no-eval-fps --no-raster-only --no-image-loss --steps 30 --warmup 5"
run() {
  rm -rf /tmp/prof_ab && mkdir -p /tmp/prof_ab
  (cd /tmp && timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o p -- python $GRAFT_REPO_ROOT/bench.py $FLAGS > /tmp/prof_ab/bench.json 2> /dev/null)
  python tools/rocprof_summary.py /tmp/prof_ab /tmp/prof_ab/sum.txt 200 > /dev/null
  echo "== $1: $(python -c "import json;d=json.loads(open('/tmp/prof_ab/bench.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'])")"
  grep -E "mlp3_fwd_kernel|mlp3_bwd_wg|expand_preprocess|expand_flags" /tmp/prof_ab/sum.txt | cut -c1-100
}
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then
      unset CGS_LIB_PATH CGS_LIB_ALLOW_STALE
      CGS_M3_TILED=0 run "rep $rep product, CGS_M3_TILED=0"
      CGS_M3_TILED=1 run "rep $rep product, CGS_M3_TILED=1"
    else
      export CGS_LIB_PATH=$GRAFT_REPO_ROOT/$lib CGS_LIB_ALLOW_STALE=1
      CGS_M3_TILED=1 run "rep $rep $lib, CGS_M3_TILED=1"
    fi
  done
done
