import os, sys
sys.path[:0]=['/root/repo','/root/repo/tests']
import torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
dev = WaveRNNDevice(synth.wavernn_state(seed=5)["model_state"])
mel = torch.from_numpy(synth.wavernn_mel(100, seed=13) / 4.0).cuda()
res={}
for mode in (None, "exact", "0"):
    if mode is None: os.environ.pop("MBHIP_WAVERNN_RESIDENT", None)
    else: os.environ["MBHIP_WAVERNN_RESIDENT"]=mode
    for r in range(3):
        s = dev.generate_samples(mel, False, 0, 0, seed=21); torch.cuda.synchronize()
    res[mode]=s.clone()
    print(mode, dev.last_path, dev.last_loop_launches, "us/step", dev.last_loop_ms*1e3/s.shape[1], "equal_to_default", bool(torch.equal(s, res[None])))
import types
from mockingbird_amd.vocoder.wavernn import hparams as hp
hpm = types.SimpleNamespace(**{k: getattr(hp, k) for k in dir(hp) if not k.startswith("_")}); hpm.voc_mode="MOL"
mol = WaveRNNDevice(synth.wavernn_state(synth.WAVERNN_HP_MOL, seed=6)["model_state"], hpm)
for mode in (None, "0"):
    if mode is None: os.environ.pop("MBHIP_WAVERNN_RESIDENT", None)
    else: os.environ["MBHIP_WAVERNN_RESIDENT"]=mode
    for r in range(2):
        s = mol.generate_samples(mel, False, 0, 0, seed=21); torch.cuda.synchronize()
    print("MOL", mode, mol.last_path, mol.last_loop_launches, "us/step", mol.last_loop_ms*1e3/s.shape[1], float(s.abs().max()))
