#!/bin/bash
for d in "" devbuf_round=0; do
echo "== MBHIP_DIAG=$d"
MBHIP_DIAG=$d python - <<'PY'
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
g1 = GanGenerator(h, st, 0, dtype="f32")
print("fresh", t(g1), t(g1))
w = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
mel = torch.from_numpy(synth.wavernn_mel(1000, seed=100) / 4.0).cuda()
w.generate_samples(mel, True, 8000, 800, seed=1); torch.cuda.synchronize()
g2 = GanGenerator(h, st, 0, dtype="f32")
print("second handle", t(g2), t(g2), "first again", t(g1))
del g1
g3 = GanGenerator(h, st, 0, dtype="f32")
print("third handle (first freed)", t(g3), t(g3))
PY
MBHIP_DIAG=$d python bench.py --no-cpu-baseline --no-tacotron --no-ppg2mel --no-wavernn-batch --no-wavernn-unbatched --no-wavernn-mol --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('bench hifigan', r['hifigan']['ms_per_batch'], r['hifigan_f16']['ms_per_batch'])"
done
