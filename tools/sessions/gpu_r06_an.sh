#!/bin/bash
python - <<'PY'
import sys, time, os
sys.path[:0] = ['.', 'tests']
import torch, synth
from mockingbird_amd.vocoder.gan import GanGenerator
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
h = synth.HIFIGAN_16K
st = synth.gan_state(h, "hifigan", seed=3)["generator"]
gm = torch.from_numpy(synth.mel_input(200, 32, seed=0)).cuda()
def t(gen, reps=5):
    for _ in range(2): gen(gm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): y = gen(gm)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 3)
w = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
mel = torch.from_numpy(synth.wavernn_mel(1000, seed=100) / 4.0).cuda()
t0 = time.time()
for _ in range(10): w.generate_samples(mel, True, 8000, 800, seed=1)
torch.cuda.synchronize(); print("wavernn phase s", time.time() - t0)
g = GanGenerator(h, st, 0, dtype="f32")
print("gan after long wavernn phase:", [t(g) for _ in range(4)])
time.sleep(0.5)
print("after 3 s idle:", [t(g) for _ in range(4)])
PY
