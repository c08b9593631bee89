#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_conv_split_tm_gpu.py tests/test_resblock_pair_split_gpu.py tests/test_gan_gpu.py -x -q 2>&1 | tail -3
python tools/taco_gen_time.py 2>&1 | tail -2
python tools/gan_run.py hifigan f32 32 200 5 2>&1 | tail -1
for t in 1 2 3; do MBHIP_DIAG=ctm_tile=$t python tools/gan_run.py hifigan f32 32 200 5 2>&1 | tail -1; done
SPAIR_NOLEG=1 SPAIR_DS=1 timeout 600 python tools/spair_bench.py 10 2>&1 | grep pair_us | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'C' in r: print('C%d k%d d%d: %.1f us  %.3f' % (r['C'], r['k'], r['d'], r['pair_us'], r['frac_833']))
    else: print(r)
"
