#!/bin/bash
# Round-5 session B: the whole WaveRNN test file, the batch-32 loop against round 4's library and two build variants, wall-clock marks of
# the headline kernel.
exec < /dev/null
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_wavernn_gpu.py -m gpu -q --timeout=600 -s > gpurun_out/r05_pytest_wavernn_b.log 2>&1; echo "pytest wavernn rc=$?"
grep -v "amdgpu.ids" gpurun_out/r05_pytest_wavernn_b.log | grep -i "passed\|failed\|error\|replay\]" | tail -12
for v in r04 default ts3_norange ts3_oneacc r04 default; do
  if [ $v = default ]; then unset MBHIP_LIB; else export MBHIP_LIB=$PWD/build_variants/libmbhip_$v.so; fi
  timeout 300 python tools/wrn_batch32_ab.py ts3,ts3 > gpurun_out/r05_b32_$v.log 2>&1
  echo "$v: $(grep 'us per step' gpurun_out/r05_b32_$v.log | tr '\n' ' ')"
done
unset MBHIP_LIB
WQ_AB_CASES=configs1_23_folds timeout 300 python tools/wrn_pipe_ab.py gpurun_out/r05_wrn_pipe_marks.json > gpurun_out/r05_pipe_marks.log 2>&1; echo "marks rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_wrn_pipe_marks.json"))["cases"]["configs1_23_folds"]
print({k: v for k, v in d.items() if k != "pipe_marks_us"})
for role, rows in d.get("pipe_marks_us", {}).items():
    for st, r in enumerate(rows[:2]):
        print(role, st, [None if x is None else round(x, 2) for x in r])
PY
