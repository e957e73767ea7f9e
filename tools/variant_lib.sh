#!/bin/bash
# tools/variant_lib.sh <tag> <source.hip[,source2.hip,...]> <extra flags...>: a variant of libcgs_hip.so in which the named
# translation units are recompiled with extra defines (timing experiments), linked with the product build's other objects.
# Run in the authoring container (hipcc cross-compiles); the result tools/variants/libcgs_<tag>.so travels to the GPU box,
# where CGS_LIB_PATH=tools/variants/libcgs_<tag>.so selects it (contextgs_amd/_lib.py).
set -e
cd "$(dirname "$0")/.."
tag=$1; srcs=$2; shift 2
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p tools/variants
others=$(ls contextgs_amd/csrc/build/*.o)
objs=""
for src in ${srcs//,/ }; do
  obj=tools/variants/$(basename $src).$tag.o
  extra=""
  if [ "$(basename $src)" = api.hip ]; then   # cgs_build_info(): same source digest as the product build, the variant's flags
    extra="-DCGS_SOURCE_DIGEST=\"$(python -c 'from contextgs_amd import build; print(build.source_digest())')\" -DCGS_BUILD_FLAGS=\"variant:$tag\""
  fi
  hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude -Icontextgs_amd/csrc -DCGS_EXPERIMENTS "$@" $extra -c contextgs_amd/csrc/$src -o $obj
  others=$(echo "$others" | grep -v "/$(basename $src).o")
  objs="$objs $obj"
done
hipcc -shared -fPIC --offload-arch=gfx950 $others $objs -o tools/variants/libcgs_$tag.so
echo tools/variants/libcgs_$tag.so
