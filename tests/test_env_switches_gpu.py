"""Every run-time switch of libmbhip that selects another code path for the SAME arithmetic is exercised here (or in
the test named beside it in DESIGN.md's switch table): a switch that is not tested does not stay in the library."""
import numpy as np
import pytest
import torch

import hiputil
import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wavernn(cuda, lib):
    from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
    return WaveRNNDevice(synth.wavernn_state(seed=5)["model_state"])


@pytest.mark.parametrize("env", [{"MBHIP_WAVERNN_CHAIN": "split"}, {"MBHIP_GRAPH_STEPS": "32"}, {"MBHIP_NO_GRAPH": "1"},
                                 {"MBHIP_WAVERNN_CHAIN": "classic"}],
                         ids=lambda e: ",".join(f"{k[6:]}={v}" for k, v in e.items()))
def test_wavernn_switches_keep_the_sample_stream(wavernn, monkeypatch, env):
    """The launch chains behind the resident kernels (MBHIP_WAVERNN_RESIDENT=0): FM fast chain (default) == round-1 row-tile chain
    == classic chain, any graph length, eager launches: the same Philox stream and bit-identical sums -> identical samples."""
    mel = torch.from_numpy(synth.wavernn_mel(45, seed=8) / 4.0).cuda()  # 3 folds of 4400: > 16 columns never; nta = 1
    mel2 = torch.from_numpy(synth.wavernn_mel(330, seed=9) / 4.0).cuda()  # 18 folds -> two column tiles
    monkeypatch.setenv("MBHIP_WAVERNN_RESIDENT", "0")
    for k in list(env):
        monkeypatch.delenv(k, raising=False)
    base = [wavernn.generate_samples(mel, True, 4000, 200, seed=11), wavernn.generate_samples(mel2, True, 4000, 400, seed=12)]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    alt = [wavernn.generate_samples(mel, True, 4000, 200, seed=11), wavernn.generate_samples(mel2, True, 4000, 400, seed=12)]
    for a, b in zip(base, alt):
        assert a.shape == b.shape
        if env.get("MBHIP_WAVERNN_CHAIN") == "classic":
            # the classic chain multiplies W_ih1 . (Ipre + x W_I[:,0]) inside the loop, the others read the pre-composed T1
            # table: the same algebra, other roundings -- identical streams except where a near-tie flips a sample (after which
            # that fold diverges, it is autoregressive)
            same = [bool(torch.equal(a[i], b[i])) for i in range(a.shape[0])]
            first_bad = [int((a[i] != b[i]).nonzero()[0]) if not same[i] else a.shape[1] for i in range(a.shape[0])]
            assert sum(same) >= a.shape[0] - 2 and min(first_bad) >= 50, (same, first_bad)
        else:
            assert torch.equal(a, b), int((a != b).sum())


@pytest.fixture(scope="module")
def taco(cuda, lib):
    from mockingbird_amd.synthesizer.inference import TacotronDevice
    st = synth.tacotron_state(seed=3)["model_state"]
    dev = TacotronDevice(st, torch.device("cuda"))
    seqs, emb = synth.tacotron_inputs(20, 30, 45, seed=6)
    T = max(len(s) for s in seqs)
    chars = torch.tensor(np.stack([np.pad(s, (0, T - len(s))) for s in seqs])).long().cuda()
    spk = torch.tensor(np.stack(emb)).cuda()
    mem, memp = dev.encode(chars, spk, -1, None, 1)
    return dev, mem, memp, chars


@pytest.mark.parametrize("env", [{"MBHIP_GRAPH_STEPS": "4"}, {"MBHIP_NO_GRAPH": "1"}],
                         ids=lambda e: ",".join(f"{k[6:]}={v}" for k, v in e.items()))
def test_tacotron_fast_loop_switches_are_bit_identical(taco, monkeypatch, env):
    """How many iterations a graph holds, eager launches: the same kernels on the same operands."""
    dev, mem, memp, chars = taco
    for k in list(env):
        monkeypatch.delenv(k, raising=False)
    base = dev.decode(mem, memp, chars, 90, 11.0, seed=5)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    alt = dev.decode(mem, memp, chars, 90, 11.0, seed=5)
    for a, b in zip(base, alt):
        assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("env,exact", [({"MBHIP_PPG_FAST": "0"}, False), ({"MBHIP_GRAPH_STEPS": "4"}, True),
                                       ({"MBHIP_NO_GRAPH": "1"}, True)], ids=["PPG_FAST=0", "GRAPH_STEPS=4", "NO_GRAPH"])
@pytest.mark.parametrize("B", [1, 19])
def test_ppg2mel_switches(cuda, lib, monkeypatch, env, exact, B):
    """The 6-launch FM step against the 8-launch general step (different summation orders: tolerance) and against
    itself with another graph length / eager launches (bit-identical)."""
    from mockingbird_amd.ppg2mel import Ppg2MelDecoder
    dec = Ppg2MelDecoder(synth.ppg2mel_decoder_state(synth.PPG2MEL_HP, seed=3, stop_bias=-6.0), synth.PPG2MEL_HP)
    mem = torch.from_numpy(synth.ppg2mel_memory(B, 20, seed=2)).cuda()
    masks = synth.ppg2mel_dropout_masks(4, 40, B)
    for k in list(env):
        monkeypatch.delenv(k, raising=False)
    base = dec.decode(mem, dropout=masks)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    alt = dec.decode(mem, dropout=masks)
    for a, b in zip(base, alt):
        assert a.shape == b.shape
        if exact:
            assert torch.equal(a, b)
        else:
            e = hiputil.relerr(a, b)
            assert e["nan"] == 0 and e["max_abs"] <= 2e-4, e


def test_wavernn_persistent_kernel_keeps_the_sample_stream(wavernn, monkeypatch):
    """wavernn_persist.h (one fold column, batched=False): ONE launch for the whole utterance, weights resident in LDS, layers handing
    their vectors over through tagged granules, products as fmaf chains in the fp32 MFMA's own order -- against the 5-launch chain:
    the same samples, sample for sample."""
    mel = torch.from_numpy(synth.wavernn_mel(9, seed=13) / 4.0).cuda()
    monkeypatch.setenv("MBHIP_WAVERNN_RESIDENT", "0")
    base = wavernn.generate_samples(mel, False, 0, 0, seed=21)
    assert wavernn.last_loop_launches > 1
    monkeypatch.setenv("MBHIP_WAVERNN_RESIDENT", "exact")  # (the default for one column is the operand-pair kernel since round 5: one group of one column)
    alt = wavernn.generate_samples(mel, False, 0, 0, seed=21)
    assert wavernn.last_loop_launches == 1 and wavernn.last_path == "persist1", "the persistent kernel did not run"
    assert base.shape == alt.shape and base.shape[0] == 1
    assert torch.equal(base, alt), (int((base != alt).sum()), int((base != alt).any(0).nonzero()[0]) if (base != alt).any() else -1)
    monkeypatch.delenv("MBHIP_WAVERNN_RESIDENT")
    dflt = wavernn.generate_samples(mel, False, 0, 0, seed=21)
    assert wavernn.last_loop_launches == 1 and wavernn.last_path == "pipe16" and dflt.shape == base.shape
    assert float((dflt == base).float().mean()) > 0.99  # same noise, fp32-grade sums: equal up to near-ties (held to the oracle in test_wavernn_gpu.py)


def test_wavernn_persistent_kernel_falls_back_to_the_chain(cuda, lib, monkeypatch):
    """When the persistent launch gives up (its workgroups were not all resident: a hand-off never arrives), the same
    call runs the launch chain -- same samples -- and the handle stops defaulting to the persistent kernel."""
    from mockingbird_amd.vocoder.wavernn.inference import WaveRNNDevice
    dev = WaveRNNDevice(synth.wavernn_state(seed=5)["model_state"])
    mel = torch.from_numpy(synth.wavernn_mel(9, seed=13) / 4.0).cuda()
    monkeypatch.setenv("MBHIP_WAVERNN_RESIDENT", "0")
    base = dev.generate_samples(mel, False, 0, 0, seed=4)
    monkeypatch.setenv("MBHIP_WAVERNN_RESIDENT", "exact")
    monkeypatch.setenv("MBHIP_DIAG", "abort_wp")
    alt = dev.generate_samples(mel, False, 0, 0, seed=4)
    assert dev.last_loop_launches > 1 and torch.equal(base, alt) and dev.last_path == "chain" and dev.last_fallback == "abort"
    monkeypatch.delenv("MBHIP_DIAG")
    again = dev.generate_samples(mel, False, 0, 0, seed=4)  # (the test switch does not mark the device as failed)
    assert dev.last_loop_launches == 1 and torch.equal(base, again) and dev.last_fallback is None
    # the pipelined resident kernel (wavernn_pipe.h) takes the same way out
    mel3 = torch.from_numpy(synth.wavernn_mel(40, seed=13) / 4.0).cuda()
    monkeypatch.setenv("MBHIP_WAVERNN_RESIDENT", "0")
    base3 = dev.generate_samples(mel3, True, 3000, 100, seed=4)
    monkeypatch.setenv("MBHIP_WAVERNN_RESIDENT", "1")
    monkeypatch.setenv("MBHIP_DIAG", "abort_wp")
    alt3 = dev.generate_samples(mel3, True, 3000, 100, seed=4)
    assert dev.last_loop_launches > 1 and torch.equal(base3, alt3)


@pytest.mark.parametrize("frames,target,overlap,folds,groups", [(40, 4000, 200, 2, 2), (40, 3000, 100, 3, 2), (330, 4000, 400, 15, 2),
                                                                (1000, 8000, 800, 23, 2), (330, 2000, 100, 32, 2), (40, 3000, 100, 3, 1),
                                                                (200, 3000, 300, 13, 1)],
                         ids=["2-folds", "3-folds", "15-folds", "23-folds", "32-folds", "3-folds-1-group", "13-folds-1-group"])
def test_wavernn_pipe_kernel_keeps_the_sample_stream(wavernn, monkeypatch, frames, target, overlap, folds, groups):
    """wavernn_pipe.h (the EXACT resident kernel: MBHIP_WAVERNN_RESIDENT=exact; also the MOL path): ONE launch of role-specialised resident
    workgroups, two fold-column groups in flight (one with MBHIP_DIAG=wq_groups=1) -- against the 5-launch chain: the same samples,
    sample for sample, from 2 to 32 columns (BASELINE configs[1] = 23).  The default kernel since round 4 (wavernn_pipe16.h, 22-bit
    operand pairs) is not bit-identical to the chain; it is held to the oracle in test_wavernn_gpu.py::test_production_*."""
    mel = torch.from_numpy(synth.wavernn_mel(frames, seed=17) / 4.0).cuda()
    monkeypatch.setenv("MBHIP_WAVERNN_RESIDENT", "0")
    base = wavernn.generate_samples(mel, True, target, overlap, seed=31)
    assert base.shape[0] == folds and wavernn.last_loop_launches == 5 * base.shape[1]
    monkeypatch.setenv("MBHIP_WAVERNN_RESIDENT", "exact")
    if groups == 1:
        monkeypatch.setenv("MBHIP_DIAG", "wq_groups=1")
    alt = wavernn.generate_samples(mel, True, target, overlap, seed=31)
    assert wavernn.last_loop_launches == 1, "the resident kernel did not run"
    bad = (base != alt)
    assert torch.equal(base, alt), (int(bad.sum()), int(bad.any(0).nonzero()[0]) if bad.any() else -1, bad.any(1).nonzero().flatten().tolist())


@pytest.mark.parametrize("name,value", [("MBHIP_WAVERNN_RESIDENT", "on"), ("MBHIP_WAVERNN_CHAIN", "clasic"), ("MBHIP_WAVERNN_CHAIN", "split,nofused")])
def test_misspelt_switch_values_fail_loudly(wavernn, monkeypatch, name, value):
    """A path switch with a value outside its documented set fails the call (MB_EINVAL with the variable's name in the message)
    instead of silently running the default path."""
    from mockingbird_amd import _lib
    mel = torch.from_numpy(synth.wavernn_mel(40, seed=1) / 4.0).cuda()
    monkeypatch.setenv(name, value)
    with pytest.raises(_lib.MbHipError) as ei:
        wavernn.generate_samples(mel, True, 3000, 100, seed=1)
    assert name in str(ei.value)
    monkeypatch.delenv(name)
    assert wavernn.generate_samples(mel, True, 3000, 100, seed=1).shape[0] == 3  # the handle is usable afterwards


def test_misspelt_wide_form_fails_loudly(wavernn, monkeypatch):
    from mockingbird_amd import _lib
    mels = [torch.from_numpy(synth.wavernn_mel(30, seed=u) / 4.0).cuda() for u in range(2)]
    monkeypatch.setenv("MBHIP_RNN_WIDE", "ts4")
    with pytest.raises(_lib.MbHipError) as ei:
        wavernn.generate_samples_batch(mels, 2000, 200, [1, 2])
    assert "MBHIP_RNN_WIDE" in str(ei.value)
    monkeypatch.setenv("MBHIP_RNN_WIDE", "ts2:2")
    assert len(wavernn.generate_samples_batch(mels, 2000, 200, [1, 2])) == 2
