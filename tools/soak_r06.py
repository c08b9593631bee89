"""Determinism soak of the round-6 kernels: the same forward many times, every result bit-identical to the first (a lost barrier or a
window overwritten too early shows up as a rare mismatch).  python tools/soak_r06.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch, synth
from mockingbird_amd.vocoder.gan import GanGenerator
from mockingbird_amd.synthesizer.inference import TacotronDevice
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = 0
for kind, h, B, F in (("hifigan", synth.HIFIGAN_16K, 32, 200), ("fregan", synth.FREGAN_16K, 8, 300), ("hifigan", synth.HIFIGAN_16K, 3, 37)):
    gen = GanGenerator(h, synth.gan_state(h, kind, seed=3)["generator"], 0 if kind == "hifigan" else 1, dtype="f32")
    mel = torch.from_numpy(synth.mel_input(F, B, seed=0)).cuda()
    ref = gen(mel).clone()
    n = 0
    for r in range(reps):
        y = gen(mel)
        if not torch.equal(y, ref):
            n += 1
    torch.cuda.synchronize()
    print(f"{kind} f32 {B}x{F}: {n} of {reps} forwards differ from the first", flush=True)
    bad += n
tst = synth.tacotron_state(seed=3)["model_state"]
seqs, emb = synth.tacotron_inputs(32, 90, 110, seed=2)
T = max(len(s) for s in seqs)
chars = torch.tensor(np.stack([np.pad(s, (0, T - len(s))) for s in seqs])).long().cuda()
spk = torch.tensor(np.stack(emb)).cuda()
dev = TacotronDevice(tst, torch.device("cuda"))
m0, l0, a0 = dev.generate(chars, spk, steps=400, style_idx=-1, min_stop_token=11, seed=5)
m0, l0 = m0.clone(), l0.clone()
n = 0
for r in range(max(4, reps // 5)):
    m, l, a = dev.generate(chars, spk, steps=400, style_idx=-1, min_stop_token=11, seed=5)
    if not (torch.equal(m, m0) and torch.equal(l, l0)):
        n += 1
print(f"tacotron generate 32 x 400: {n} of {max(4, reps // 5)} runs differ from the first")
bad += n
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
