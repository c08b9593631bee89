#!/bin/bash
# Diagnostics: link libmbhip with ONE module (default rnn; MODULE=resblock_f16 ...) rebuilt under extra -D flags into
# build_variants/libmbhip_<name>.so (select at run time with MBHIP_LIB=...).
# usage: [MODULE=rnn] tools/build_variant.sh <name> [-DFLAG ...]
set -eu
name=$1; shift
mod=${MODULE:-rnn}
cd "$(dirname "$0")/.."
mkdir -p build_variants
C=mockingbird_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -Wno-unused-result "$@" -c $C/$mod.hip -o build_variants/${mod}_$name.o
objs=$(ls $C/*.o | grep -v "/$mod\.o\$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build_variants/libmbhip_$name.so $objs build_variants/${mod}_$name.o
rm -f build_variants/${mod}_$name.o
echo "built build_variants/libmbhip_$name.so"
