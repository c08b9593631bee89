#!/bin/bash
# round 6: the whole GPU suite + smoke + the bench line (tag = $1)
TAG=${1:-a}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_pytest_gpu_$TAG.log; cat gpurun_out/r06_pytest_gpu_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke_$TAG.log 2>&1; tail -4 gpurun_out/r06_smoke_$TAG.log
python bench.py 2>/dev/null | tail -1 > gpurun_out/r06_bench_$TAG.log; python - <<PY
import json
r=json.load(open('gpurun_out/r06_bench_$TAG.log'))
print('value', r['value'], r['config']['us_per_time_step'], 'roofline', r['roofline']['frac'])
for k in ('hifigan','hifigan_f16','fregan_f16'): print(k, r[k]['ms_per_batch'], r[k]['roofline']['frac'])
t=r['tacotron']; print('taco', t['ms_per_batch'], t['us_per_decoder_iteration'], t['postnet_ms'], t['encoder_ms'])
print('batch32', r['wavernn_batch32'].get('us_per_time_step'), 'e2e', r['e2e_configs3']['value'], 'ppg', r['ppg2mel']['batch1'].get('us_per_step'), r['ppg2mel']['batch32'].get('us_per_step'))
PY
