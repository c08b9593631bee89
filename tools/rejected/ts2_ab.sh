#!/bin/bash
# A/B of the wide-batch recurrent GEMM forms (rnn_body.h TS vs rnn_ts2_body.h): bit-exact batch tests, then the
# batch-32 throughput object of bench.py with each form.
set -u
mkdir -p gpurun_out
for v in 1 0; do
  echo "== MBHIP_RNN_TS2=$v"
  MBHIP_RNN_TS2=$v timeout 300 python -m pytest tests/test_wavernn_gpu.py -m gpu -q -k "batch" 2>&1 | tail -3
  MBHIP_RNN_TS2=$v timeout 300 python bench.py --no-hifigan --no-tacotron --no-ppg2mel --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null | tail -1 > gpurun_out/ts2_ab_$v.json
  python - <<PY
import json
r = json.load(open("gpurun_out/ts2_ab_$v.json"))
b = r["wavernn_batch32"]
print("ts2=$v batch32: %.0f samples/s, %.1f us/step, loop %.1f ms; N=1 value %.0f" % (b["value"], b["us_per_time_step"], b["sample_loop_ms"], r["value"]))
PY
done
