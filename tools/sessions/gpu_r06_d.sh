#!/bin/bash
# round 6 session D: the pipelined support waves (parity, per-shape timing, forward time)
mkdir -p gpurun_out
python -m pytest tests/test_resblock_pair_split_gpu.py tests/test_gan_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r06_d_pytest.log
cat gpurun_out/r06_d_pytest.log
SPAIR_NOLEG=1 timeout 600 python tools/spair_bench.py 10 > gpurun_out/r06_d_spair_bench.log 2>&1
grep pair_us gpurun_out/r06_d_spair_bench.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'C' in r: print('C%d k%d d%d: %.1f us  %.3f' % (r['C'], r['k'], r['d'], r['pair_us'], r['frac_833']))
    else: print(r)
"
for t in 1 2; do echo "== spair_tile=$t"; MBHIP_DIAG=spair_tile=$t SPAIR_NOLEG=1 SPAIR_DS=1 timeout 600 python tools/spair_bench.py 10 2>&1 | grep pair_us | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'C' in r: print('C%d k%d d%d: %.1f us  %.3f' % (r['C'], r['k'], r['d'], r['pair_us'], r['frac_833']))
"; done
python tools/gan_run.py hifigan f32 32 200 5 2>&1 | tail -1
python tools/gan_run.py fregan f32 8 1000 3 2>&1 | tail -1
