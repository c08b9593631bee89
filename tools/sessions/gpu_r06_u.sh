#!/bin/bash
python -m pytest tests/test_resblock_pair_split_gpu.py tests/test_conv_split_tm_gpu.py tests/test_gan_gpu.py -x -q 2>&1 | tail -3
for lib in "" notapw; do
echo "== lib $lib"
( [ -n "$lib" ] && export MBHIP_LIB=$PWD/build_variants/libmbhip_$lib.so; SPAIR_NOLEG=1 SPAIR_DS=1 timeout 600 python tools/spair_bench.py 10 64,20000 32,40000 2>&1 | grep pair_us | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'C' in r: print('C%d k%d d%d: %.1f us  %.3f' % (r['C'], r['k'], r['d'], r['pair_us'], r['frac_833']))
    else: print(r)
"; python tools/gan_run.py hifigan f32 32 200 5 2>&1 | tail -1 )
done
for sh in "64 20000 3 1" "32 40000 3 1" "32 40000 11 1"; do
  echo "== $sh"
  MBHIP_LIB=$PWD/build_variants/libmbhip_sptrace.so python tools/spair_trace.py $sh 2>&1 | grep -v amdgpu.ids | tail -3
done
