#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gan_rb2_gpu.py tests/test_pipeline_gpu.py tests/test_gan_gpu.py tests/test_vits_gpu.py tests/test_conv_split_tm_gpu.py tests/test_resblock_pair_split_gpu.py -q -s 2>&1 | grep -v "^$" | tail -15 | tee gpurun_out/r06_m_pytest_a.log
python bench.py --no-wavernn-unbatched --no-wavernn-mol --no-ppg2mel --steps 3 --warmup 1 2>/dev/null | tail -1 > gpurun_out/r06_m_bench.json
python - <<'PY'
import json
r=json.load(open('gpurun_out/r06_m_bench.json'))
print(list(r.keys()))
print('value', r['value'], 'config', {k:v for k,v in r['config'].items() if 'exact' in k or 'us_per' in k})
print('roofline exact', {k:v for k,v in r['roofline'].items() if 'exact' in k or k in ('frac','achieved')})
for k in ('hifigan','hifigan_f16','fregan_f16'):
    if k in r: print(k, r[k].get('ms_per_batch'), r[k]['roofline'].get('frac'))
t=r.get('tacotron',{})
print('taco', {k:t.get(k) for k in ('ms_per_batch','decoder_loop_ms','postnet_ms','encoder_ms','us_per_decoder_iteration')}, t.get('postnet_roofline'))
print('e2e', r.get('e2e_configs3',{}).get('value'))
PY
