#!/bin/bash
# Wave-state / MFMA counters of the wide-batch WaveRNN loop kernels (eager launches: counter mode + hipGraph crash)
export TMPDIR=/tmp MBHIP_NO_GRAPH=1
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_b
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_b -o p -- python bench.py --no-cpu-baseline --no-hifigan --no-tacotron --no-ppg2mel --steps 1 --warmup 0 > gpurun_out/pmc_b.log 2>&1
echo "rc=$?"
python tools/pmc_summary.py gpurun_out/pmc_b gpurun_out/pmc_batch_summary.json | grep -E "_ts_kernel" | head
rm -rf gpurun_out/pmc_b
