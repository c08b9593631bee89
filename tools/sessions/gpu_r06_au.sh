#!/bin/bash
python -m pytest tests/test_pipeline_gpu.py tests/test_tacotron_gpu.py -x -q -m gpu 2>&1 | tail -4
python bench.py --no-cpu-baseline --no-ppg2mel --no-wavernn-batch --no-wavernn-unbatched --no-wavernn-mol 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); e=r['e2e_configs3']; print('e2e', e['value'], e['s_total'], e['shares_rank0']['synthesizer_s'], e['shares_rank0']['vocoder_s'], e['shares_rank0']['gather_s'])"
