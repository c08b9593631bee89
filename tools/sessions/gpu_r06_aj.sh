#!/bin/bash
# round 6: why does bench.py's hifigan object read 9.2 ms when tools/gan_run.py reads 8.0-8.2?
for reps in 5 20 5; do python tools/gan_run.py hifigan f32 32 200 $reps 2>&1 | tail -1; done
python - <<'PY'
import sys, time
sys.path[:0] = ['.', 'tests']
import torch, synth
from mockingbird_amd.vocoder.gan import GanGenerator
h = synth.HIFIGAN_16K
st = synth.gan_state(h, "hifigan", seed=3)["generator"]
gm = torch.from_numpy(synth.mel_input(200, 32, seed=0)).cuda()
gen = GanGenerator(h, st, 0, dtype="f32")
def t(reps):
    for _ in range(2): gen(gm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): y = gen(gm)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for r in (5, 5, 20, 5): print("bench-style reps", r, t(r))
PY
python bench.py --no-cpu-baseline --no-tacotron --no-ppg2mel --no-wavernn-batch --no-wavernn-unbatched --no-wavernn-mol --no-e2e 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('bench hifigan', r['hifigan']['ms_per_batch'], r['hifigan_f16']['ms_per_batch'])"
