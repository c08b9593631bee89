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
gen = GanGenerator(h, st, 0, dtype="f32")
def t(reps=5):
    for _ in range(2): gen(gm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): y = gen(gm)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("before wavernn", t(), t())
w = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
mel = torch.from_numpy(synth.wavernn_mel(1000, seed=100) / 4.0).cuda()
for _ in range(3): w.generate_samples(mel, True, 8000, 800, seed=1)
torch.cuda.synchronize()
print("after wavernn", t(), t())
gen2 = GanGenerator(h, st, 0, dtype="f32")
gen = gen2
print("new handle after wavernn", t(), t())
del w; torch.cuda.empty_cache()
print("after freeing wavernn", t(), t())
PY
