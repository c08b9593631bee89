#!/bin/bash
exec < /dev/null
mkdir -p gpurun_out/r04h
timeout 600 python -m pytest tests/test_wavernn_gpu.py -x -q -m gpu -k "batch" > gpurun_out/r04h/pytest_batch.log 2>&1; tail -4 gpurun_out/r04h/pytest_batch.log
timeout 300 python tools/wrn_batch32_ab.py > gpurun_out/r04h/batch32_ab.log 2>&1; tail -4 gpurun_out/r04h/batch32_ab.log
cat > /tmp/b32.py <<'PY'
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
dev = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
mels = [torch.from_numpy(synth.wavernn_mel(1000, seed=100 + u) / 4.0).cuda() for u in range(32)]
outs = dev.generate_samples_batch(mels, 8000, 800, list(range(32)))
torch.cuda.synchronize()
print(dev.last_batch_plan.n_folds, outs[0].shape, dev.last_loop_ms * 1e3 / outs[0].shape[1])
PY
MBHIP_NO_GRAPH=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04h/prof -o p -- python /tmp/b32.py > gpurun_out/r04h/prof.log 2>&1
grep -v "^W2\|^E2\|^I2" gpurun_out/r04h/prof.log | tail -2
f=$(find gpurun_out/r04h/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" gpurun_out/r04h/kernel_stats_batch32_ts3.csv; head -8 "$f" | cut -c1-200; fi
rm -rf gpurun_out/r04h/prof
