#!/bin/bash
# per-kernel durations of the wide WaveRNN batch loop (736 columns, eager launches): rnn_ts3_body.h against rnn_ts2_body.h
exec < /dev/null
mkdir -p gpurun_out/r04g
cat > /tmp/b32.py <<'PY'
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
dev = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
mels = [torch.from_numpy(synth.wavernn_mel(120, seed=100 + u) / 4.0).cuda() for u in range(32)]
outs = dev.generate_samples_batch(mels, 800, 80, list(range(32)))
torch.cuda.synchronize()
print(dev.last_batch_plan.n_folds, outs[0].shape, dev.last_loop_ms * 1e3 / outs[0].shape[1])
PY
for d in ts3 ts2; do
  if [ $d = ts2 ]; then export MBHIP_RNN_WIDE=ts2; fi
  MBHIP_NO_GRAPH=1 timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04g/prof_$d -o p -- python /tmp/b32.py > gpurun_out/r04g/prof_$d.log 2>&1
  grep -v "^W2\|^E2\|^I2" gpurun_out/r04g/prof_$d.log | tail -2
  f=$(find gpurun_out/r04g/prof_$d -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" gpurun_out/r04g/kernel_stats_$d.csv; head -9 "$f" | cut -c1-220; fi
  rm -rf gpurun_out/r04g/prof_$d
done
