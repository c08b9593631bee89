#!/bin/bash
echo "== default"; bash tools/sessions/gpu_r06_an.sh 2>&1 | tail -2
echo "== gan_one_stream"; MBHIP_DIAG=gan_one_stream bash tools/sessions/gpu_r06_an.sh 2>&1 | tail -2
python tools/gan_run.py hifigan f32 32 200 20 2>&1 | tail -1
python bench.py --no-cpu-baseline --no-tacotron --no-ppg2mel --no-wavernn-batch --no-wavernn-unbatched --no-wavernn-mol --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('bench value', r['value'], 'hifigan', r['hifigan']['ms_per_batch'], r['hifigan_f16']['ms_per_batch'])"
python -m pytest tests/test_wavernn_gpu.py tests/test_env_switches_gpu.py tests/test_resident_stress_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu 2>&1 | tail -3
