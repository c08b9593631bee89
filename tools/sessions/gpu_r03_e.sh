#!/bin/bash
# GPU session E of round 3: store-shape changes of the resident kernels (one store per workgroup and hand-off): ppg2mel and
# WaveRNN tests, A/B timings, unbatched WaveRNN loop time
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ppg2mel_gpu.py tests/test_wavernn_gpu.py tests/test_env_switches_gpu.py -m gpu -q -x --timeout=300 > gpurun_out/pytest_e.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_e.log
MBHIP_DIAG=pr_trace=/tmp/pr_trace.bin timeout 300 python tools/ppg_resident_ab.py > gpurun_out/ppg_resident_ab.log 2>&1; echo "ab rc=$?"
grep T_enc gpurun_out/ppg_resident_ab.log
timeout 600 python tools/wrn_pipe_ab.py > gpurun_out/pipe_ab.log 2>&1; echo "pipe_ab rc=$?"
tail -c 1500 gpurun_out/pipe_ab.log
timeout 300 python - <<'PY'
import os, sys, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests")]
import torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
dev = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
mel = torch.from_numpy(synth.wavernn_mel(100, seed=0) / 4.0).cuda()
for rep in range(3):
    s = dev.generate_samples(mel, False, 0, 0, seed=1)
    torch.cuda.synchronize()
    print("unbatched: steps", s.shape[1], "us/step", dev.last_loop_ms * 1e3 / s.shape[1], "launches", dev.last_loop_launches)
PY
