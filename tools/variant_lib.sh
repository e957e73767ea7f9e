#!/bin/bash
# tools/variant_lib.sh <tag> <source.hip> <extra flags...>: a variant of libcgs_hip.so in which ONE translation unit is
# recompiled with extra defines (timing experiments), linked with the product build's other objects.  Run in the authoring
# container (hipcc cross-compiles); the result tools/variants/libcgs_<tag>.so travels to the GPU box, where
# CGS_LIB_PATH=tools/variants/libcgs_<tag>.so selects it (contextgs_amd/_lib.py).
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift 2
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p tools/variants
obj=tools/variants/$(basename $src).$tag.o
hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Iinclude -Icontextgs_amd/csrc -DCGS_EXPERIMENTS "$@" -c contextgs_amd/csrc/$src -o $obj
others=$(ls contextgs_amd/csrc/build/*.o | grep -v "/$(basename $src).o")
hipcc -shared -fPIC --offload-arch=gfx950 $others $obj -o tools/variants/libcgs_$tag.so
echo tools/variants/libcgs_$tag.so
