#!/bin/bash
# Round-5 session C: the production WaveRNN tests, headline A/B (round 4's library, this tree), marks, the batch-32 loop.
exec < /dev/null
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-c}
timeout 900 python -m pytest tests/test_wavernn_gpu.py -m gpu -q --timeout=600 -s -k "production or mol or resident" > gpurun_out/r05_pytest_wavernn_$TAG.log 2>&1; echo "pytest wavernn rc=$?"
grep -v "amdgpu.ids" gpurun_out/r05_pytest_wavernn_$TAG.log | grep -i "passed\|failed\|error\|replay\]" | tail -12
MBHIP_LIB=$PWD/build_variants/libmbhip_r04.so timeout 300 python tools/wrn_ab_r05.py r04_$TAG quick > gpurun_out/r05_ab_r04_$TAG.log 2>&1
timeout 300 python tools/wrn_ab_r05.py r05_$TAG > gpurun_out/r05_ab_r05_$TAG.log 2>&1; echo "ab r05 rc=$?"
for v in ${VARIANTS:-}; do
  MBHIP_LIB=$PWD/build_variants/libmbhip_$v.so timeout 300 python tools/wrn_ab_r05.py ${v}_$TAG quick > gpurun_out/r05_ab_${v}_$TAG.log 2>&1
  echo "variant $v: $(grep -h '^configs1_default' gpurun_out/r05_ab_${v}_$TAG.log | cut -c1-120)"
done
grep -h "^configs1\|folds" gpurun_out/r05_ab_r04_$TAG.log gpurun_out/r05_ab_r05_$TAG.log | cut -c1-200
WQ_AB_CASES=configs1_23_folds timeout 300 python tools/wrn_pipe_ab.py gpurun_out/r05_wrn_pipe_marks_$TAG.json > gpurun_out/r05_pipe_marks_$TAG.log 2>&1; echo "marks rc=$?"
python - <<PY
import json
d = json.load(open("gpurun_out/r05_wrn_pipe_marks_$TAG.json"))["cases"]["configs1_23_folds"]
print({k: v for k, v in d.items() if k not in ("pipe_marks_us", "publish_us_step1001")})
for role, rows in d.get("pipe_marks_us", {}).items():
    for st, r in enumerate(rows[:2]):
        print(role, st, [None if x is None else round(x, 2) for x in r])
for role, xs in d.get("publish_us_step1001", {}).items():
    v = [x for x in xs if x is not None]
    if v: print("publish", role, "min", min(v), "max", max(v), "first", xs[0], "last", xs[-1])
PY
timeout 300 python tools/wrn_batch32_ab.py ts3,ts3 2>&1 | grep "us per step"
