"""Split (error-compensated fp16 MFMA) vs fp32-input MFMA conv kernel on the bench objects that run fp32 convs:
HiFi-GAN fp32 B=32x200 and Tacotron generate B=32.  usage: python tools/conv_split_bench.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch, synth
from mockingbird_amd.vocoder.gan import GanGenerator
out = {}
h = synth.HIFIGAN_16K
st = synth.gan_state(h, "hifigan", seed=6)["generator"]
mel = torch.from_numpy(synth.mel_input(200, 32, seed=1)).cuda()
ys = {}
for mode in ("split", "f32"):
    os.environ["MBHIP_CONV_SPLIT"] = "1" if mode == "split" else "0"
    g = GanGenerator(h, st, 0, dtype="f32")
    y = g(mel); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        y = g(mel)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    ys[mode] = y
    out["hifigan_f32_32x200_" + mode] = {"ms": ms, "tflops": 352.1e6 * 200 * 32 / (ms * 1e-3) / 1e12}
d = (ys["split"] - ys["f32"]).float()
out["hifigan_split_vs_f32"] = {"rms": float(d.pow(2).mean().sqrt()), "max": float(d.abs().max()), "ref_rms": float(ys["f32"].pow(2).mean().sqrt())}
from mockingbird_amd.synthesizer.inference import TacotronDevice
tst = synth.tacotron_state(seed=3)["model_state"]
seqs, emb = synth.tacotron_inputs(32, 90, 110, seed=2)
T = max(len(s) for s in seqs)
chars = torch.tensor(np.stack([np.pad(s, (0, T - len(s))) for s in seqs])).long().cuda()
spk = torch.tensor(np.stack(emb)).cuda()
ms_ = {}
for mode in ("split", "f32"):
    os.environ["MBHIP_CONV_SPLIT"] = "1" if mode == "split" else "0"
    dev = TacotronDevice(tst, torch.device("cuda"))
    r = dev.generate(chars, spk, steps=400, style_idx=-1, min_stop_token=11, seed=5); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r = dev.generate(chars, spk, steps=400, style_idx=-1, min_stop_token=11, seed=5)
    torch.cuda.synchronize()
    ms_[mode] = ((time.perf_counter() - t0) / 3 * 1e3, r)
    out["tacotron_generate_b32_" + mode] = {"ms": ms_[mode][0]}
dm = (ms_["split"][1][1] - ms_["f32"][1][1]).abs().max()
out["tacotron_split_vs_f32_linear_max_abs"] = float(dm)
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "conv_split_bench.json"), "w"), indent=1)
