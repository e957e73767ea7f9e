cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_rate_sub_gpu.py -x -q -s 2>&1 | tail -30) > gpurun_out/r06_t1.log
rm -rf /tmp/prof_m
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o m -- python tools/rate_sub_micro.py ${MICRO_ARGS:-807417 121000 71} > gpurun_out/r06_micro_prof.log 2>&1
python tools/rocprof_summary.py /tmp/prof_m gpurun_out/r06_micro_prof.txt 20 | cut -c1-200
cat gpurun_out/r06_t1.log | tail -12
