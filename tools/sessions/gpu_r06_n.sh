#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_conv_split_tm_gpu.py -x -q 2>&1 | tail -3
python -m pytest tests/test_tacotron_gpu.py -x -q -s 2>&1 | grep -v "^$" | tail -25 | tee gpurun_out/r06_n_pytest_taco.log
python tools/taco_gen_time.py 2>&1 | tail -2
MBHIP_DIAG=taco_post_cm python tools/taco_gen_time.py 2>&1 | tail -2
python -m pytest tests/test_ppg2mel_gpu.py -x -q -s -k small 2>&1 | grep -v "^$" | tail -5
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_n -o taco -- python tools/taco_gen_time.py > /dev/null 2>&1
f=$(find gpurun_out/prof_n -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/r06_n_taco_kernel_stats.csv; rm -rf gpurun_out/prof_n
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r06_n_taco_kernel_stats.csv')))
for r in rows[:28]:
    n=int(r['Calls']); t=float(r['TotalDurationNs'])
    print(f"{r['Name'][:100]:100s} calls/gen {n/6:7.1f}  us/gen {t/6/1e3:8.1f}  avg {float(r['AverageNs'])/1e3:7.1f}")
PY
