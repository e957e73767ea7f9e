#!/bin/bash
# tools/variants_prof.sh <kernel-regex> <tag>...: rocprofv3 kernel times of a short headline bench run for the product library
# and for each tools/variants/libcgs_<tag>.so (built by tools/variant_lib.sh in the authoring container)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
pat=$1; shift
for tag in product "$@"; do
  if [ $tag = product ]; then unset CGS_LIB_PATH; else export CGS_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/libcgs_$tag.so; fi
  rm -rf /tmp/vp; mkdir -p /tmp/vp
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vp -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-raster-only --no-codec --no-image-loss --no-heavy --no-eval-fps > /tmp/vp/bench.json 2>/dev/null)
  python tools/rocprof_summary.py /tmp/vp /tmp/vp.txt 60 > /dev/null
  echo "== $tag: $(python -c "import json;d=json.load(open('/tmp/vp/bench.json'));print(d['value'], d['ms_per_step'])")"
  grep -E "$pat" /tmp/vp.txt | cut -c1-130
done
