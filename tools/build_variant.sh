#!/bin/bash
# build_variants/libst_<name>.so = the library with one source recompiled with extra flags (same-box A/B through ST_KERNELS_LIB)
#   tools/build_variant.sh <name> <source.hip> [-DFLAG ...]
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/.."
python -c "from style_transfer_amd import build; build.build(verbose=False)"
mkdir -p build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c -o build_variants/$name.o style_transfer_amd/csrc/$src
objs=$(ls style_transfer_amd/build/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o build_variants/libst_$name.so $objs build_variants/$name.o
echo build_variants/libst_$name.so
