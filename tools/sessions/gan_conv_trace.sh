#!/bin/bash
# per-shape conv kernel times of one fp32 HiFi-GAN forward (B=32 x 200 frames): bash tools/gan_conv_trace.sh
export TMPDIR=/tmp
rm -rf gpurun_out/prof_gan32
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gan32 -o g -- python tools/gan_run.py hifigan f32 32 200 1 > gpurun_out/gan32.log 2>&1
tail -1 gpurun_out/gan32.log
f=$(find gpurun_out/prof_gan32 -name "*kernel_trace.csv" | head -1)
python - "$f" <<PY
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
conv = [r for r in rows if "conv1d" in r["Kernel_Name"]]
half = conv[len(conv)//2:]
acc = collections.OrderedDict()
for r in half:
    key = (r["Kernel_Name"][25:62], r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = acc.setdefault(key, [0, 0.0, []]); a[0] += 1; a[1] += d; a[2].append(round(d))
tot = 0
for k, (n, t, ds) in acc.items():
    print(k, n, round(t, 1), "us total,", round(t / n, 1), "avg", ds if n > 1 else ""); tot += t
print("total conv us", round(tot, 1))
PY
rm -rf gpurun_out/prof_gan32
