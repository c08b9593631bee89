#!/bin/bash
# round 4, session B: conv split (scaled residual) parity + the headline kernel's A/B switches (MBHIP_DIAG=wq_flags) + marks
mkdir -p gpurun_out/r04b
python -m pytest tests/test_conv1d_gpu.py tests/test_gan_gpu.py -x -q -m gpu -s > gpurun_out/r04b/pytest_conv.log 2>&1; tail -3 gpurun_out/r04b/pytest_conv.log
python tools/wrn_pipe_sweep.py 1,5,9,17,33,65,129,13,29,61,157,189,221,253 > gpurun_out/r04b/wq_sweep.log 2>&1; cat gpurun_out/r04b/wq_sweep.log
cp gpurun_out/wq_sweep.json gpurun_out/r04b/ 2>/dev/null
WQ_AB_CASES=configs1_23_folds MBHIP_DIAG=wq_flags=1 python tools/wrn_pipe_ab.py gpurun_out/r04b/ab_flags1.json > gpurun_out/r04b/ab_flags1.log 2>&1
WQ_AB_CASES=configs1_23_folds MBHIP_DIAG=wq_flags=253 python tools/wrn_pipe_ab.py gpurun_out/r04b/ab_flags253.json > gpurun_out/r04b/ab_flags253.log 2>&1
tail -2 gpurun_out/r04b/ab_flags253.log | cut -c1-600
python tools/gan_run.py hifigan f32 32 200 5 > gpurun_out/r04b/hifigan_f32.log 2>&1; tail -3 gpurun_out/r04b/hifigan_f32.log
