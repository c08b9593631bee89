#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_conv_split_tm_gpu.py tests/test_gan_gpu.py -x -q 2>&1 | tail -3
python -m pytest tests/test_tacotron_gpu.py -x -q 2>&1 | tail -3
python tools/taco_gen_time.py 2>&1 | tail -2
python tools/gan_run.py hifigan f32 32 200 5 2>&1 | tail -1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_p -o taco -- python tools/taco_gen_time.py > /dev/null 2>&1
f=$(find gpurun_out/prof_p -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'gru_scan_kernel<8' in r['Kernel_Name']][-1]
for r in rows[idx-17:idx+4]:
    d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    g=r.get('Grid_Size') or r.get('Grid_Size_X') or '?'
    print(f"{r['Kernel_Name'][:70]:70s} grid {g:>8s}  {d:8.1f} us")
PY
rm -rf gpurun_out/prof_p
