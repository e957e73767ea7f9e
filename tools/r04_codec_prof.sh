cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf /tmp/prof_codec
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_codec -o c -- python $GRAFT_REPO_ROOT/tools/codec_v2_prof.py ${1:-2} > /dev/null 2>&1)
python tools/rocprof_summary.py /tmp/prof_codec gpurun_out/r04_codec_v${1:-2}_kernels.txt 40 > /dev/null; head -34 gpurun_out/r04_codec_v${1:-2}_kernels.txt | cut -c1-150
