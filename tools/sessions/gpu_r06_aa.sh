#!/bin/bash
# round 6: window loads through range-checked buffer descriptors (no select at the request): tests, A/B against the previous library, traces
python -m pytest tests/test_conv_split_tm_gpu.py tests/test_resblock_pair_split_gpu.py -x -q -m gpu 2>&1 | tail -3
for r in 1 2; do
for lib in mockingbird_amd/libmbhip.so build_variants/libmbhip_head.so; do
  echo "== $lib"
  MBHIP_LIB=$lib python tools/gan_run.py hifigan f32 32 200 20 2>&1 | tail -1
  [ $r = 1 ] && MBHIP_LIB=$lib python tools/gan_run.py fregan f32 8 1000 10 2>&1 | tail -1
  [ $r = 1 ] && MBHIP_LIB=$lib python tools/taco_gen_time.py 2>&1 | tail -2
done; done
export MBHIP_LIB=build_variants/libmbhip_cttrace.so
for a in "64 64 20000 3" "32 1 40000 7" "256 640 1000 3"; do echo "== ctm $a"; python tools/ctm_trace.py $a 2>&1 | tail -4; done
export MBHIP_LIB=build_variants/libmbhip_sptrace.so
for a in "64 25600 3 1" "64 25600 11 5" "32 51200 3 1" "256 1600 3 1" "128 12800 7 3"; do
  echo "== $a"; python tools/spair_trace.py $a 2>&1 | tail -6
done
