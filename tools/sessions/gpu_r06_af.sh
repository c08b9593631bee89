#!/bin/bash
# round 6: 8 against 4 support waves at <= 64 channels now that the window loads fly through the interval
export SPAIR_NOLEG=1 SPAIR_DS=1
for lib in mockingbird_amd/libmbhip.so build_variants/libmbhip_nl4.so mockingbird_amd/libmbhip.so build_variants/libmbhip_nl4.so; do
  echo "== MBHIP_LIB=$lib"
  MBHIP_LIB=$lib python tools/spair_bench.py 10 64,20000 32,40000 2>&1 | tail -7
done
