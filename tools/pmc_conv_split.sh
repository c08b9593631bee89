export TMPDIR=/tmp
rm -rf gpurun_out/pmc5
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/pmc5 -o p -- python tools/conv_micro.py 32 64 64 20000 7 3 3 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc5 gpurun_out/pmc5_v3.json | grep split | head -2
rm -rf gpurun_out/pmc5
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc5 -o p -- python tools/conv_micro.py 32 64 64 20000 7 3 3 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/pmc5 gpurun_out/pmc5b_v3.json | grep split | head -2
rm -rf gpurun_out/pmc5
