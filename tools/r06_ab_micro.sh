# same-box A/B of library variants on the rate-subset micro-benchmark: tools/r06_ab_micro.sh tag1 tag2 ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=gpurun_out/r06_ab_micro.log
: > $out
for rep in 1 2; do
for tag in base "$@"; do
  if [ $tag = base ]; then unset CGS_LIB_PATH; else export CGS_LIB_PATH=tools/variants/libcgs_$tag.so; fi
  echo "== $tag (rep $rep)" >> $out
  timeout 300 python tools/rate_sub_micro.py 2>&1 | grep "fused" >> $out
done
done
cat $out
