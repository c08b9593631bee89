"""MOL-mode WaveRNN: the resident pipelined kernel against the 5-launch chain on BASELINE configs[1]'s geometry (mel 80x1000,
23 folds x 9600 steps), same seed: per-step times and whether the two streams are bit-identical."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, synth
from mockingbird_amd.vocoder.wavernn import hparams as hp
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
st = synth.wavernn_state(synth.WAVERNN_HP_MOL, seed=6)
hpm = types.SimpleNamespace(**{k: getattr(hp, k) for k in dir(hp) if not k.startswith("_")})
hpm.voc_mode = "MOL"
dev = WaveRNNDevice(st["model_state"], hpm)
mel = torch.from_numpy(synth.wavernn_mel(1000, seed=0) / 4.0).cuda()
dev.generate_samples(mel[:, :60], True, 2000, 200, seed=1)
a = dev.generate_samples(mel, True, 8000, 800, seed=2); torch.cuda.synchronize()
print("pipe : launches", dev.last_loop_launches, "us/step %.2f" % (dev.last_loop_ms * 1e3 / a.shape[1]))
os.environ["MBHIP_WAVERNN_RESIDENT"] = "0"
c = dev.generate_samples(mel, True, 8000, 800, seed=2); torch.cuda.synchronize()
print("chain: launches", dev.last_loop_launches, "us/step %.2f" % (dev.last_loop_ms * 1e3 / c.shape[1]), "identical", bool(torch.equal(a, c)),
      "max |diff| %.3g" % float((a - c).abs().max()), "first differing step", int(((a != c).any(0)).float().argmax()) if not torch.equal(a, c) else -1)
