#!/bin/bash
# round 6: shader-clock traces of the ResBlock-unit kernel at the four HiFi-GAN stage shapes (diagnostics build)
export MBHIP_LIB=build_variants/libmbhip_sptrace.so
for a in "64 25600 3 1" "64 25600 7 3" "64 25600 11 5" "32 51200 3 1" "32 51200 11 5" "256 1600 3 1" "256 1600 11 5" "128 12800 7 3"; do
  echo "== $a"; python tools/spair_trace.py $a 2>&1 | tail -9
done
