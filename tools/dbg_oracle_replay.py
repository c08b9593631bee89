import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, synth
from oracle import wavernn as ow
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
st = synth.wavernn_state(seed=5)
dev = WaveRNNDevice(st["model_state"]); w = dict(st["model_state"])
frames, target, overlap, steps, seed = 1000, 8000, 800, 2000, 1234
mel = synth.wavernn_mel(frames, seed=1)
s = dev.generate_samples(torch.from_numpy(mel / 4.0).cuda(), True, target, overlap, seed=seed).cpu()
noise = dev.sampler_noise(seed, steps, 23).cpu()
print("noise min", float(noise.min()), "zeros", int((noise == 0).sum()), "max", float(noise.max()))
torch.set_num_threads(8)
with torch.no_grad():
    mels, aux = ow.conditioning(w, ow.HP, torch.from_numpy(mel[None] / 4.0), True, target, overlap)
    o_s, o_l = ow.sample_loop(w, ow.HP, mels, aux, noise=noise, forced=s, return_logits=True, max_steps=steps)
k_dev = torch.round((s[:, :steps] + 1) * 511 / 2).long(); k_or = torch.round((o_s + 1) * 511 / 2).long()
mism = (k_dev != k_or).nonzero()
print("mismatches", mism.tolist())
for n, t in mism.tolist():
    l = o_l[t, n].double(); E = noise[t, n].double()
    key = l - torch.log(E)
    top = key.topk(3)
    post = torch.softmax(o_l[t, n], 0) / noise[t, n]
    print("fold", n, "step", t, "dev", int(k_dev[n, t]), "oracle", int(k_or[n, t]), "fp64 gumbel top3", top.indices.tolist(), top.values.tolist(),
          "E at dev/oracle class", float(E[k_dev[n, t]]), float(E[k_or[n, t]]), "post top2", post.topk(2).values.tolist(), post.topk(2).indices.tolist(),
          "logits", float(l[k_dev[n, t]]), float(l[k_or[n, t]]), "lmax", float(l.max()))
