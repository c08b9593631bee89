"""GAN vocoder passes for profiling: python tools/gan_run.py [hifigan|fregan] [f32|f16] [batch] [frames] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, synth
from mockingbird_amd.vocoder.gan import GanGenerator
kind = sys.argv[1] if len(sys.argv) > 1 else "hifigan"
dtype = sys.argv[2] if len(sys.argv) > 2 else "f16"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
F = int(sys.argv[4]) if len(sys.argv) > 4 else 200
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
h = synth.HIFIGAN_16K if kind == "hifigan" else synth.FREGAN_16K
gen = GanGenerator(h, synth.gan_state(h, kind, seed=3)["generator"], 0 if kind == "hifigan" else 1, dtype=dtype)
mel = torch.from_numpy(synth.mel_input(F, B, seed=0)).cuda()
gen(mel); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    y = gen(mel)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
mflop = 352.1 if kind == "hifigan" else 385.1
print(f"{kind} {dtype} B={B} F={F}: {ms:.3f} ms/batch, {B*F*200/ms*1e3/16000:.0f}x RT, {mflop*1e6*F*B/ms/1e9:.1f} TFLOP/s")
