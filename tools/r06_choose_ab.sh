# cgs_ctx_choose_flags with its closing atomics spread over slots (CGS_CHOOSE_SLOTS): tests, then kernel times under rocprofv3, same box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_ctx_ops_gpu.py tests/test_context_gpu.py tests/test_training_gpu.py tests/test_training_parity_gpu.py tests/test_rate_sub_gpu.py -x -q 2>&1 | tail -2)
FLAGS="--no-cpu-baseline --no-codec --no-heavy --no-eval-fps --no-raster-only --no-image-loss --steps 30 --warmup 5"
for rep in 1 2; do for s in 32 1 8 64; do
  rm -rf /tmp/prof_ab && mkdir -p /tmp/prof_ab
  (cd /tmp && CGS_CHOOSE_SLOTS=$s timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o p -- python $GRAFT_REPO_ROOT/bench.py $FLAGS > /tmp/prof_ab/bench.json 2> /dev/null)
  python tools/rocprof_summary.py /tmp/prof_ab /tmp/prof_ab/sum.txt 200 > /dev/null
  echo "== rep $rep slots $s: $(grep -E "ctx_choose_flags|ctx_choose_compact" /tmp/prof_ab/sum.txt | awk '{printf "%s us  ", $3}')"
done; done
