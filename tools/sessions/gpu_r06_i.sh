#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_resblock_pair_split_gpu.py -x -q 2>&1 | tail -3
SPAIR_NOLEG=1 SPAIR_DS=1 timeout 600 python tools/spair_bench.py 10 2>&1 | grep pair_us | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'C' in r: print('C%d k%d d%d: %.1f us  %.3f' % (r['C'], r['k'], r['d'], r['pair_us'], r['frac_833']))
    else: print(r)
" | tee gpurun_out/r06_i_spair_bench.log
for sh in "256 1000 11 1" "128 5000 11 1" "64 20000 3 1" "64 20000 11 1" "32 40000 11 1"; do
  echo "== $sh"
  MBHIP_LIB=$PWD/build_variants/libmbhip_sptrace.so python tools/spair_trace.py $sh 2>&1 | grep -v amdgpu.ids | tail -3
done | tee gpurun_out/r06_i_trace.log
python tools/gan_run.py hifigan f32 32 200 5 2>&1 | tail -1
python tools/gan_run.py fregan f32 8 1000 3 2>&1 | tail -1
