#!/bin/bash
# Diagnostics: link libmbhip with rnn.hip rebuilt under extra -D flags into build_variants/libmbhip_<name>.so
# (select at run time with MBHIP_LIB=...).  usage: tools/build_variant.sh <name> [-DFLAG ...]
set -eu
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p build_variants
C=mockingbird_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -Wno-unused-result "$@" -c $C/rnn.hip -o build_variants/rnn_$name.o
objs=$(ls $C/*.o | grep -v '/rnn\.o$')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build_variants/libmbhip_$name.so $objs build_variants/rnn_$name.o
rm -f build_variants/rnn_$name.o
echo "built build_variants/libmbhip_$name.so"
