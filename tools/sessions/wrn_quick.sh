#!/bin/bash
# WaveRNN quick check: parity tests of the loop + headline numbers (us/step batched and unbatched, per-kernel marginals)
timeout 900 python -m pytest tests/test_wavernn_gpu.py -q -x 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-hifigan --no-tacotron --no-ppg2mel --no-wavernn-batch --no-e2e 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('samples/s', round(r['value']), 'us/step', round(r['config']['us_per_time_step'],2), 'unbatched us/step', round(r.get('wavernn_unbatched',{}).get('us_per_time_step',0),2))
print({k:round(v['avg_us'],2) for k,v in r['roofline']['per_kernel'].items()}, 'frac', round(r['roofline']['frac'],4), 'whole-step GB/s', round(r['roofline']['whole_step']['GBps']))"
