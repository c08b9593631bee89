#!/bin/bash
# round 6: shader-clock traces of conv_split_tm at the HiFi-GAN upsampler / conv_post shapes (diagnostics build)
export MBHIP_LIB=build_variants/libmbhip_cttrace.so
for a in "64 64 20000 3" "32 1 40000 7" "256 640 1000 3" "128 256 5000 3" "512 1280 200 3"; do echo "== ctm $a"; python tools/ctm_trace.py $a 2>&1 | tail -9; done
