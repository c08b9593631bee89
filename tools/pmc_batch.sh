#!/bin/bash
# Wave-state / MFMA counters of the wide-batch WaveRNN loop kernels (eager launches: counter mode + hipGraph crash)
export TMPDIR=/tmp MBHIP_NO_GRAPH=1
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_b
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_b -o p -- python bench.py --no-cpu-baseline --no-hifigan --no-tacotron --no-ppg2mel --steps 1 --warmup 0 > gpurun_out/pmc_b.log 2>&1
echo "rc=$?"
python tools/pmc_summary.py gpurun_out/pmc_b gpurun_out/pmc_batch_summary.json | grep -E "_ts2?_kernel" | head
python - <<'PY'
# mean kernel duration (counter mode serialises dispatches, durations stay per-kernel) -> shader clock estimate
import csv, glob, json, collections
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("gpurun_out/pmc_b/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        a = dur[row["Kernel_Name"]]; a[0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); a[1] += 1
s = json.load(open("gpurun_out/pmc_batch_summary.json"))
for k, (t, n) in dur.items():
    if "_ts" in k and k in s:
        us = t / n / 1e3
        c = s[k]
        gui = c.get("GRBM_GUI_ACTIVE", {}).get("mean_per_dispatch", 0) / 8
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("mean_per_dispatch", 0)
        s[k]["mean_duration_us"] = us
        print(k[:70], "dur %.1f us, clock %.2f GHz, mfma_busy/gui %.3f" % (us, gui / us / 1e3, mf / max(gui, 1) / 1024.0))
json.dump(s, open("gpurun_out/pmc_batch_summary.json", "w"), indent=1)
PY
rm -rf gpurun_out/pmc_b
