#!/bin/bash
# round 6 session A: first light of the split pair kernel (parity tests, per-shape timing against the per-conv launches)
mkdir -p gpurun_out
python -m pytest tests/test_resblock_pair_split_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r06_a_pytest.log
cat gpurun_out/r06_a_pytest.log
timeout 600 python tools/spair_bench.py 10 > gpurun_out/r06_a_spair_bench.log 2>&1
cat gpurun_out/r06_a_spair_bench.log
