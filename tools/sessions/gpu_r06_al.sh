#!/bin/bash
python - <<'PY'
import sys, time, os
sys.path[:0] = ['.', 'tests']
import torch, synth
from mockingbird_amd import _lib
from mockingbird_amd.vocoder.gan import GanGenerator
h = synth.HIFIGAN_16K
st = synth.gan_state(h, "hifigan", seed=3)["generator"]
gm = torch.from_numpy(synth.mel_input(200, 32, seed=0)).cuda()
gen = GanGenerator(h, st, 0, dtype="f32")
L = _lib.lib()
need = L.mb_gan_workspace_bytes(gen._h, 32, 200)
big = torch.empty(need + (64 << 20), dtype=torch.uint8, device="cuda")
print("need MB", need / 2**20, "big base %x" % big.data_ptr())
def t(reps=8):
    for _ in range(2): gen(gm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): y = gen(gm)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for off in (0, 512, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 3 << 20, 0):
    gen._ws = big[off:off + need]
    print("offset", off, "ms", round(t(), 3))
PY
