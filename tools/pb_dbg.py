import os, sys
ROOT='/root/repo'
sys.path[:0]=[ROOT, ROOT+'/tests']
import torch, synth, hiputil
from oracle import ppg2mel as op
from mockingbird_amd.ppg2mel import Ppg2MelDecoder
for B in (2, 16):
    T=24
    w = synth.ppg2mel_decoder_state(synth.PPG2MEL_HP, seed=3, stop_bias=0.0)
    dec = Ppg2MelDecoder(w, synth.PPG2MEL_HP)
    mem = torch.from_numpy(synth.ppg2mel_memory(B, T, seed=3))
    masks = synth.ppg2mel_dropout_masks(11, T*2, B)
    with torch.no_grad():
        omel, oal, ostop = op.inference_batched(w, dict(op.HP), mem, masks=op.MaskSource(list(masks)))
    for mode, diag in (("0", None), ("1", None), ("1", "pb_groups=2")):
        os.environ["MBHIP_PPG_RESIDENT"]=mode
        if diag: os.environ["MBHIP_DIAG"]=diag
        else: os.environ.pop("MBHIP_DIAG", None)
        mel, al, stop = dec.decode(mem.cuda(), dropout=masks)
        n=min(al.shape[1], oal.shape[1])
        m=mel.cpu().reshape(B,-1,80)
        d=(m[:, :2*n]-omel[:, :2*n]).abs()
        per_step=[float(d[:, 2*k:2*k+2].max()) for k in range(min(n,8))]
        print(B, mode, diag, "steps", al.shape[1], oal.shape[1], "launches", dec.last_loop_launches, "mel", float(d.max()), "align", float((al.cpu()[:, :n]-oal[:, :n]).abs().max()), "first steps", ["%.1e"%x for x in per_step])
