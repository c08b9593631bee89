"""Stress of the resident (one-launch) loops' hand-offs -- wf_pipe16_kernel (23 fold columns, RAW and MOL models), wf_pipe_kernel (the exact kernel),
wf_persist1_kernel (one column), ppg_resident_kernel (ppg2mel, one utterance), ppg_batch_kernel (ppg2mel, 32 utterances), taco_front_kernel (the
Tacotron iteration's fused front: a launch with two in-launch hand-offs per iteration, counted as "1" here against the 7-launch iteration): the same call many times, alone and while another
stream keeps the GPU busy with GEMMs of varying size (uneven load, workgroups competing for compute units).  Every run must either
reproduce the quiet resident run bit for bit (the kernels are deterministic) or -- if the launch lost a hand-off and drained -- equal
the launch chain's result (the fallback); anything else is a FAILURE.  VERDICT r03 item 8.
usage: python tools/resident_stress.py [reps]   (prints one line per run, then 'ok' or 'FAILED')"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch, synth
from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
from mockingbird_amd.ppg2mel import Ppg2MelDecoder
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 9
dev = WaveRNNDevice(synth.wavernn_state(seed=1)["model_state"])
import types
from mockingbird_amd.vocoder.wavernn import hparams as _whp
_hpm = types.SimpleNamespace(**{k: getattr(_whp, k) for k in dir(_whp) if not k.startswith("_")})
_hpm.voc_mode = "MOL"
dev_mol = WaveRNNDevice(synth.wavernn_state(synth.WAVERNN_HP_MOL, seed=6)["model_state"], _hpm)
dec = Ppg2MelDecoder(synth.ppg2mel_decoder_state(synth.PPG2MEL_HP, seed=3, stop_bias=-6.0), synth.PPG2MEL_HP)
mel23 = torch.from_numpy(synth.wavernn_mel(120, seed=0) / 4.0).cuda()   # 23 x 1100-step folds at target 1000 / overlap 50? (see columns below)
mel1 = torch.from_numpy(synth.wavernn_mel(12, seed=0) / 4.0).cuda()
mem = torch.from_numpy(synth.ppg2mel_memory(1, 60, seed=2)).cuda()
mem32 = torch.from_numpy(synth.ppg2mel_memory(32, 60, seed=3)).cuda()   # ppg_batch.h: 160 role workgroups + 32 attention workgroups


def wrn(mel, batched, env, d=None):
    d = d or dev
    for k in ("MBHIP_WAVERNN_RESIDENT",):
        os.environ.pop(k, None)
    os.environ.update(env)
    out = d.generate_samples(mel, batched, 1000, 50, seed=9)
    torch.cuda.synchronize()
    return out.clone(), d.last_loop_launches


from mockingbird_amd.synthesizer.inference import TacotronDevice
import numpy as np
_tst = synth.tacotron_state(seed=3)["model_state"]
tdev = TacotronDevice(_tst, torch.device("cuda"))
_seqs, _emb = synth.tacotron_inputs(32, 60, 100, seed=4)
_T = max(len(q) for q in _seqs)
t_chars = torch.tensor(np.stack([np.pad(q, (0, _T - len(q))) for q in _seqs])).long().cuda()
t_mem, t_memp = tdev.encode(t_chars, torch.tensor(np.stack(_emb)).cuda(), -1, None, 1)


def taco(env):
    os.environ.pop("MBHIP_DIAG", None)
    os.environ.update(env)
    mel, _, _ = tdev.decode(t_mem, t_memp, t_chars, 120, 11.0, seed=3)
    torch.cuda.synchronize()
    os.environ.pop("MBHIP_DIAG", None)
    return mel.clone(), (1 if tdev.last_loop_launches_per_iteration in (4, 5) else 7)


def ppg(env, m=None):
    os.environ.pop("MBHIP_PPG_RESIDENT", None)
    os.environ.update(env)
    out = dec.decode(mem if m is None else m, seed=5, max_steps=100)
    torch.cuda.synchronize()
    return out[0].clone(), dec.last_loop_launches


CASES = {
    "wavernn_pipe16": (lambda env: wrn(mel23, True, env), {"MBHIP_WAVERNN_RESIDENT": "1"}, {"MBHIP_WAVERNN_RESIDENT": "0"}),
    "wavernn_pipe_exact": (lambda env: wrn(mel23, True, env), {"MBHIP_WAVERNN_RESIDENT": "exact"}, {"MBHIP_WAVERNN_RESIDENT": "0"}),
    "wavernn_pipe16_mol": (lambda env: wrn(mel23, True, env, dev_mol), {"MBHIP_WAVERNN_RESIDENT": "1"}, {"MBHIP_WAVERNN_RESIDENT": "0"}),
    "wavernn_one_column": (lambda env: wrn(mel1, False, env), {"MBHIP_WAVERNN_RESIDENT": "1"}, {"MBHIP_WAVERNN_RESIDENT": "0"}),
    "ppg2mel_resident": (ppg, {"MBHIP_PPG_RESIDENT": "1"}, {"MBHIP_PPG_RESIDENT": "0"}),
    "ppg2mel_batch32_resident": (lambda env: ppg(env, mem32), {"MBHIP_PPG_RESIDENT": "1"}, {"MBHIP_PPG_RESIDENT": "0"}),
    "tacotron_fused_front_b32": (taco, {}, {"MBHIP_DIAG": "taco_front=0"}),
}
side = torch.cuda.Stream()
bad = 0
for name, (run, env_res, env_chain) in CASES.items():
    chain, nl = run(env_chain)
    assert nl > 1, (name, "chain reference ran", nl)
    quiet, nl = run(env_res)
    assert nl == 1, (name, "the resident launch did not run on the quiet GPU", nl)
    for rep in range(REPS):
        load = rep % 3  # 0: idle, 1: small GEMMs, 2: large GEMMs on the side stream
        if load:
            n = 512 if load == 1 else 4096
            a = torch.randn(n, n, device="cuda"); b = torch.randn(n, n, device="cuda")
            with torch.cuda.stream(side):
                for _ in range(1500 if load == 1 else 120):
                    a = (a @ b) * 1e-3
        t0 = time.perf_counter()
        # explicit switches: a device that fell back once is tried again (the memo only changes the DEFAULT)
        out, nl = run(env_res)
        verdict = "resident" if (nl == 1 and torch.equal(out, quiet)) else "fallback" if (nl > 1 and torch.equal(out, chain)) else "WRONG"
        bad += verdict == "WRONG"
        print(f"{name} rep {rep} load {load}: {verdict} (launches {nl}, wall {time.perf_counter() - t0:.3f} s)", flush=True)
        torch.cuda.synchronize()
print("FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
