"""Seeded synthetic checkpoints in the reference's state-dict layouts (there
are no pretrained checkpoints anywhere).  No reference import
needed: shapes come from the configs.  Scales are chosen so activations stay
O(1) (the reference's N(0,0.01) GAN init gives ~1e-6 outputs, making absolute
tolerances vacuous, SURVEY.md section 7 hard part 5)."""
import math

import numpy as np
import torch

HIFIGAN_16K = {
    "resblock": "1", "upsample_rates": [5, 5, 4, 2], "upsample_kernel_sizes": [10, 10, 8, 4],
    "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "num_mels": 80,
    "sampling_rate": 16000, "seed": 1234,
}
# The reference ships no 24 kHz json; h.sampling_rate == 24000 switches Generator to the
# Interpolate(nearest)+Conv1d upsampler (hifigan/models.py:107-118).  Rates for hop 300 (12.5 ms @ 24 kHz);
# kernels mix odd ("same" length) and even (the stage comes out one sample short).
HIFIGAN_24K = {
    "resblock": "1", "upsample_rates": [5, 5, 4, 3], "upsample_kernel_sizes": [11, 10, 8, 7],
    "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "num_mels": 80,
    "sampling_rate": 24000, "seed": 1234,
}
# h.resblock == '2' (hifigan/models.py:100): the V3-style stack, ResBlock2 units with the first two dilations of every block
HIFIGAN_RB2 = {
    "resblock": "2", "upsample_rates": [8, 8, 4], "upsample_kernel_sizes": [16, 16, 8],
    "upsample_initial_channel": 256, "resblock_kernel_sizes": [3, 5, 7],
    "resblock_dilation_sizes": [[1, 2], [2, 6], [3, 12]], "num_mels": 80, "sampling_rate": 16000, "seed": 1234,
}
GAN_RB2_CASES = ((64, 12, 2, 7), (256, 7, 1, 8))  # (upsample_initial_channel, frames, batch, seed)
GAN24K_CASES = ((64, 16, 2, 5), (256, 9, 1, 6))  # (upsample_initial_channel, frames, batch, seed)
FREGAN_16K = {
    "resblock": "1", "upsample_rates": [5, 5, 2, 2, 2], "upsample_kernel_sizes": [10, 10, 4, 4, 4],
    "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5, 7], [1, 3, 5, 7], [1, 3, 5, 7]], "num_mels": 80,
    "sampling_rate": 16000, "seed": 1234,
}


def small(cfg, uic=64):
    c = dict(cfg)
    c["upsample_initial_channel"] = uic
    return c


def _wn(rng, shape, row_norm):
    """weight_g / weight_v pair whose folded weight has per-dim0-row L2 norm `row_norm`."""
    v = rng.standard_normal(shape).astype(np.float32)
    g = np.full((shape[0],) + (1,) * (len(shape) - 1), row_norm, np.float32)
    g *= (1.0 + 0.1 * rng.standard_normal(g.shape)).astype(np.float32)
    return torch.from_numpy(g), torch.from_numpy(v)


def gan_state(h, kind="hifigan", seed=0, top_k=4):
    """{'generator': state_dict} with weight_g/weight_v/bias per conv, as
    hifigan/inference.py:47-52 expects before remove_weight_norm()."""
    rng = np.random.default_rng(seed)
    uic = h["upsample_initial_channel"]
    rates, ks = h["upsample_rates"], h["upsample_kernel_sizes"]
    sd = {}

    def conv(name, cout, cin, k, gain=1.0):  # Conv1d weight [cout][cin][k]
        g, v = _wn(rng, (cout, cin, k), gain)
        sd[name + ".weight_g"], sd[name + ".weight_v"] = g, v
        sd[name + ".bias"] = torch.from_numpy((0.05 * rng.standard_normal(cout)).astype(np.float32))

    def convT(name, cin, cout, k, u, gain=1.0):  # ConvTranspose1d weight [cin][cout][k]; norm per cin row
        # each output sums cin*(k/u) taps of variance row_norm^2/(cout*k) -> var = cin*row_norm^2/(cout*u)
        rn = gain * math.sqrt(cout * u / cin)
        g, v = _wn(rng, (cin, cout, k), rn)
        sd[name + ".weight_g"], sd[name + ".weight_v"] = g, v
        sd[name + ".bias"] = torch.from_numpy((0.05 * rng.standard_normal(cout)).astype(np.float32))

    conv("conv_pre", uic, h["num_mels"], 7, 1.0 / 1.5)
    for i, (u, k) in enumerate(zip(rates, ks)):
        if kind == "hifigan" and int(h.get("sampling_rate", 16000)) == 24000:
            conv(f"ups.{i}.1", uic >> (i + 1), uic >> i, k, 1.6)  # Sequential(Interpolate, Conv1d) models.py:107-118
        else:
            convT(f"ups.{i}", uic >> i, uic >> (i + 1), k, u, 1.3)
    if kind == "fregan":
        lvl = len(rates) - top_k
        kr = h["num_mels"]
        for i in range(lvl, len(rates)):
            convT(f"cond_up.{i - lvl}", kr, uic >> i, ks[i - 1], rates[i - 1], 0.7)
            kr = uic >> i
        for i in range(lvl + 1, len(rates)):
            conv(f"res_output.{i - lvl - 1}.1", uic >> (i + 1), uic >> i, 1, 0.7)
    nk = len(h["resblock_kernel_sizes"])
    for i in range(len(rates)):
        ch = uic >> (i + 1)
        for j, k in enumerate(h["resblock_kernel_sizes"]):
            nd = len(h["resblock_dilation_sizes"][j])
            if str(h.get("resblock", "1")) != "1":  # ResBlock2 (models.py:51-72): convs[0..1]
                for d in range(2):
                    conv(f"resblocks.{i * nk + j}.convs.{d}", ch, ch, k, 0.7)
                continue
            for d in range(nd):
                conv(f"resblocks.{i * nk + j}.convs1.{d}", ch, ch, k, 1.2)
            for d in range(nd):
                conv(f"resblocks.{i * nk + j}.convs2.{d}", ch, ch, k, 0.6)
    conv("conv_post", 1, uic >> len(rates), 7, 0.25)
    return {"generator": sd}


def mel_input(frames, batch=1, seed=0, n_mels=80):
    """clip(N(0,1.5), -4, 4) float32."""
    rng = np.random.default_rng(seed)
    m = np.clip(rng.normal(0, 1.5, (batch, n_mels, frames)), -4, 4).astype(np.float32)
    return m


WAVERNN_HP = dict(rnn_dims=512, fc_dims=512, bits=9, pad=2, upsample_factors=(5, 5, 8), feat_dims=80,
                  compute_dims=128, res_out_dims=128, res_blocks=10, hop_length=256, sample_rate=16000,
                  mode="RAW", mu_law=True, apply_preemphasis=True, preemphasis=0.97,
                  mel_max_abs_value=4.0)


def wavernn_state(hp=WAVERNN_HP, seed=0):
    """{'model_state': state_dict} in the reference WaveRNN layout (fatchord_version.py:88-122)."""
    rng = np.random.default_rng(seed)
    R, FC, A = hp["rnn_dims"], hp["fc_dims"], hp["res_out_dims"] // 4
    mol = hp.get("mode", "RAW") == "MOL"
    CD, FEAT, C = hp["compute_dims"], hp["feat_dims"], (30 if mol else 2 ** hp["bits"])
    sd = {}

    def t(shape, scale):
        return torch.from_numpy((scale * rng.standard_normal(shape)).astype(np.float32))

    def bn(p, n):
        sd[p + ".weight"] = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32))
        sd[p + ".bias"] = t((n,), 0.1)
        sd[p + ".running_mean"] = t((n,), 0.1)
        sd[p + ".running_var"] = torch.from_numpy(rng.uniform(0.5, 1.5, n).astype(np.float32))
        sd[p + ".num_batches_tracked"] = torch.tensor(1, dtype=torch.long)

    k = 2 * hp["pad"] + 1
    sd["upsample.resnet.conv_in.weight"] = t((CD, FEAT, k), 1.0 / math.sqrt(FEAT * k))
    bn("upsample.resnet.batch_norm", CD)
    for i in range(hp["res_blocks"]):
        p = f"upsample.resnet.layers.{i}."
        sd[p + "conv1.weight"] = t((CD, CD, 1), 1.0 / math.sqrt(CD))
        sd[p + "conv2.weight"] = t((CD, CD, 1), 0.5 / math.sqrt(CD))
        bn(p + "batch_norm1", CD)
        bn(p + "batch_norm2", CD)
    sd["upsample.resnet.conv_out.weight"] = t((hp["res_out_dims"], CD, 1), 0.3 / math.sqrt(CD))
    sd["upsample.resnet.conv_out.bias"] = t((hp["res_out_dims"],), 0.1)
    for i, s in enumerate(hp["upsample_factors"]):
        wk = np.full((1, 1, 1, 2 * s + 1), 1.0 / (2 * s + 1), np.float32)
        wk *= (1.0 + 0.05 * rng.standard_normal(wk.shape)).astype(np.float32)  # "trained" box filter
        sd[f"upsample.up_layers.{2 * i + 1}.weight"] = torch.from_numpy(wk)
    sd["I.weight"] = t((R, FEAT + A + 1), 1.0 / math.sqrt(FEAT + A + 1))
    sd["I.bias"] = t((R,), 0.1)
    for n, kin in (("rnn1", R), ("rnn2", R + A)):
        sd[f"{n}.weight_ih_l0"] = t((3 * R, kin), 1.0 / math.sqrt(kin))
        sd[f"{n}.weight_hh_l0"] = t((3 * R, R), 1.0 / math.sqrt(R))
        sd[f"{n}.bias_ih_l0"] = t((3 * R,), 0.1)
        sd[f"{n}.bias_hh_l0"] = t((3 * R,), 0.1)
    sd["fc1.weight"] = t((FC, R + A), 1.0 / math.sqrt(R + A)); sd["fc1.bias"] = t((FC,), 0.1)
    sd["fc2.weight"] = t((FC, FC + A), 1.4 / math.sqrt(FC + A)); sd["fc2.bias"] = t((FC,), 0.1)
    sd["fc3.weight"] = t((C, FC), 2.0 / math.sqrt(FC)); sd["fc3.bias"] = t((C,), 0.1)
    if mol:  # (logits | means | log scales) x 10: means inside [-1, 1], scales ~ exp(-3.5)
        sd["fc3.weight"][10:20] *= 0.25
        sd["fc3.weight"][20:30] *= 0.15
        sd["fc3.bias"][20:30] -= 3.5
    sd["step"] = torch.zeros(1, dtype=torch.long)
    return {"model_state": sd}


WAVERNN_HP_MOL = dict(WAVERNN_HP, mode="MOL")


def mol_uniforms(seed, steps, folds, nr_mix=10):
    """uniform_(1e-5, 1 - 1e-5) draws in the order sample_from_discretized_mix_logistic consumes them per step
    (distribution.py:105,118): [folds, nr_mix] indicator draws, then [folds] logistic draws -> [steps, folds, nr_mix + 1]."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(steps):
        t1 = torch.empty(1, folds, nr_mix).uniform_(1e-5, 1.0 - 1e-5, generator=g)
        t2 = torch.empty(1, folds).uniform_(1e-5, 1.0 - 1e-5, generator=g)
        out.append(torch.cat([t1[0], t2[0][:, None]], dim=1))
    return torch.stack(out)


def wavernn_mel(frames, seed=1, n_mels=80):
    """U(0,4) float32; infer_waveform divides by 4."""
    rng = np.random.default_rng(seed)
    return rng.uniform(0, 4, (n_mels, frames)).astype(np.float32)


def exp_noise(seed, steps, folds, classes):
    """Exp(1) draws in the order torch.multinomial consumes them: one (folds, classes) tensor per step."""
    g = torch.Generator().manual_seed(seed)
    return torch.stack([torch.empty(folds, classes).exponential_(1, generator=g) for _ in range(steps)])


def tacotron_state(seed=0, r=2, num_chars=75, n_mels=80, P_enc=256, spk=256, gst_E=512, D=128, H=1024, C=512, K=5, nh=4):
    """{'model_state': state_dict} in the reference Tacotron layout (tacotron.py:140-162 with use_gst)."""
    rng = np.random.default_rng(seed)
    sd = {}

    def t(shape, scale):
        return torch.from_numpy((scale * rng.standard_normal(shape)).astype(np.float32))

    def lin(name, out, inp, gain=1.0, bias=True):
        sd[name + ".weight"] = t((out, inp), gain / math.sqrt(inp))
        if bias:
            sd[name + ".bias"] = t((out,), 0.1)

    def bn(p, n):
        sd[p + ".weight"] = torch.from_numpy((1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32))
        sd[p + ".bias"] = t((n,), 0.1)
        sd[p + ".running_mean"] = t((n,), 0.1)
        sd[p + ".running_var"] = torch.from_numpy(rng.uniform(0.5, 1.5, n).astype(np.float32))
        sd[p + ".num_batches_tracked"] = torch.tensor(1, dtype=torch.long)

    def cbhg(p, cin, ch, proj):
        for k in range(1, K + 1):
            sd[f"{p}.conv1d_bank.{k - 1}.conv.weight"] = t((ch, cin, k), 1.4 / math.sqrt(cin * k))
            bn(f"{p}.conv1d_bank.{k - 1}.bnorm", ch)
        sd[f"{p}.conv_project1.conv.weight"] = t((proj[0], K * ch, 3), 1.4 / math.sqrt(K * ch * 3)); bn(f"{p}.conv_project1.bnorm", proj[0])
        sd[f"{p}.conv_project2.conv.weight"] = t((proj[1], proj[0], 3), 1.0 / math.sqrt(proj[0] * 3)); bn(f"{p}.conv_project2.bnorm", proj[1])
        if proj[1] != ch:
            lin(f"{p}.pre_highway", ch, proj[1], 1.0, bias=False)
        for i in range(nh):
            lin(f"{p}.highways.{i}.W1", ch, ch, 1.4)
            lin(f"{p}.highways.{i}.W2", ch, ch, 1.0)
        for sfx in ("", "_reverse"):
            sd[f"{p}.rnn.weight_ih_l0{sfx}"] = t((3 * ch // 2, ch), 1.0 / math.sqrt(ch))
            sd[f"{p}.rnn.weight_hh_l0{sfx}"] = t((3 * ch // 2, ch // 2), 1.0 / math.sqrt(ch // 2))
            sd[f"{p}.rnn.bias_ih_l0{sfx}"] = t((3 * ch // 2,), 0.1)
            sd[f"{p}.rnn.bias_hh_l0{sfx}"] = t((3 * ch // 2,), 0.1)

    P = P_enc + spk + gst_E
    sd["encoder.embedding.weight"] = t((num_chars, 512), 1.0)
    lin("encoder.pre_net.fc1", P_enc, 512, 1.4); lin("encoder.pre_net.fc2", P_enc, P_enc, 1.4)
    cbhg("encoder.cbhg", P_enc, P_enc, [P_enc, P_enc])
    lin("encoder_proj", D, P, 1.0, bias=False)
    filt = [1, 32, 32, 64, 64, 128, 128]
    for i in range(6):
        sd[f"gst.encoder.convs.{i}.weight"] = t((filt[i + 1], filt[i], 3, 3), 1.4 / math.sqrt(filt[i] * 9))
        sd[f"gst.encoder.convs.{i}.bias"] = t((filt[i + 1],), 0.1)
        bn(f"gst.encoder.bns.{i}", filt[i + 1])
    sd["gst.encoder.gru.weight_ih_l0"] = t((3 * gst_E // 2, 128 * 4), 1.0 / math.sqrt(512))
    sd["gst.encoder.gru.weight_hh_l0"] = t((3 * gst_E // 2, gst_E // 2), 1.0 / math.sqrt(gst_E // 2))
    sd["gst.encoder.gru.bias_ih_l0"] = t((3 * gst_E // 2,), 0.1); sd["gst.encoder.gru.bias_hh_l0"] = t((3 * gst_E // 2,), 0.1)
    sd["gst.stl.embed"] = t((10, gst_E // 8), 0.5)
    lin("gst.stl.attention.W_query", gst_E, gst_E // 2 + spk, 1.0, bias=False)
    lin("gst.stl.attention.W_key", gst_E, gst_E // 8, 1.0, bias=False)
    lin("gst.stl.attention.W_value", gst_E, gst_E // 8, 1.0, bias=False)
    sd["decoder.r"] = torch.tensor(r, dtype=torch.int)
    lin("decoder.prenet.fc1", 2 * D, n_mels, 1.4); lin("decoder.prenet.fc2", 2 * D, 2 * D, 1.4)
    sd["decoder.attn_net.conv.weight"] = t((32, 1, 31), 1.0 / math.sqrt(31)); sd["decoder.attn_net.conv.bias"] = t((32,), 0.1)
    lin("decoder.attn_net.L", D, 32, 1.0, bias=False); lin("decoder.attn_net.W", D, D, 1.0)
    lin("decoder.attn_net.v", 1, D, 4.0, bias=False)  # sharper attention than a unit-gain init
    sd["decoder.attn_rnn.weight_ih"] = t((3 * D, P + 2 * D), 1.0 / math.sqrt(P + 2 * D)); sd["decoder.attn_rnn.weight_hh"] = t((3 * D, D), 1.0 / math.sqrt(D))
    sd["decoder.attn_rnn.bias_ih"] = t((3 * D,), 0.1); sd["decoder.attn_rnn.bias_hh"] = t((3 * D,), 0.1)
    lin("decoder.rnn_input", H, P + D, 1.0)
    for n in ("res_rnn1", "res_rnn2"):
        sd[f"decoder.{n}.weight_ih"] = t((4 * H, H), 1.0 / math.sqrt(H)); sd[f"decoder.{n}.weight_hh"] = t((4 * H, H), 1.0 / math.sqrt(H))
        sd[f"decoder.{n}.bias_ih"] = t((4 * H,), 0.1); sd[f"decoder.{n}.bias_hh"] = t((4 * H,), 0.1)
    lin("decoder.mel_proj", n_mels * 20, H, 1.5, bias=False)
    lin("decoder.stop_proj", 1, H + P, 1.0)
    cbhg("postnet", n_mels, C, [C, n_mels])
    lin("post_proj", n_mels, C, 1.0, bias=False)
    sd["step"] = torch.zeros(1, dtype=torch.long)
    sd["stop_threshold"] = torch.tensor(-3.4, dtype=torch.float32)
    return {"model_state": sd}


def tacotron_inputs(batch, t_min=90, t_max=110, seed=2):
    """ids U{2..74}, length U{t_min..t_max}, EOS=1 appended; L2-normalised N(0,1) speaker embeds."""
    rng = np.random.default_rng(seed)
    seqs, embeds = [], []
    for _ in range(batch):
        n = int(rng.integers(t_min, t_max + 1))
        seqs.append(np.concatenate([rng.integers(2, 75, n), [1]]).astype(np.int64))
        e = rng.standard_normal(256).astype(np.float32)
        embeds.append(e / np.linalg.norm(e))
    return seqs, embeds


def decoder_dropout_masks(seed, n_iter, batch, width=256):
    """Bernoulli(0.5) keep masks [n_iter, 2, batch, width] (pre_net.py:23,26 order within an iteration)."""
    g = torch.Generator().manual_seed(seed)
    return torch.empty(n_iter, 2, batch, width).bernoulli_(0.5, generator=g)


PPG2MEL_HP = dict(enc_dim=256, num_mels=80, frames_per_step=2, attention_rnn_dim=512, decoder_rnn_dim=512,
                  prenet_dims=(256, 128), num_mixtures=5, encoder_down_factor=4, num_decoder_rnn_layer=1,
                  concat_context_to_last=True)


def ppg2mel_decoder_state(hp=PPG2MEL_HP, seed=0, stop_bias=-2.0):
    """state_dict of models/ppg2mel/rnn_decoder_mol.py:Decoder (names and shapes as the reference module
    registers them) with O(1) activations.  MoL biases as MOLAttention.initialize_bias sets them
    (sigma 1.0; Delta -0.432 for r = frames_per_step / encoder_down_factor = 0.5)."""
    rng = np.random.default_rng(seed)
    E, nm, r = hp["enc_dim"], hp["num_mels"], hp["frames_per_step"]
    A, D, M = hp["attention_rnn_dim"], hp["decoder_rnn_dim"], hp["num_mixtures"]

    def mat(rows, cols, gain=1.0):
        return torch.from_numpy((gain * rng.standard_normal((rows, cols)) / np.sqrt(cols)).astype(np.float32))

    def vec(n, scale=0.05):
        return torch.from_numpy((scale * rng.standard_normal(n)).astype(np.float32))

    sd = {}
    dims = [nm] + list(hp["prenet_dims"])
    for name in ("prenet", "prenet_pitch"):  # prenet_pitch is registered but unused at inference
        for i in range(len(dims) - 1):
            sd[f"{name}.layers.{i}.linear_layer.weight"] = mat(dims[i + 1], dims[i], 1.4)
    P = dims[-1]
    sd["attention_rnn.weight_ih"] = mat(4 * A, P + E); sd["attention_rnn.weight_hh"] = mat(4 * A, A)
    sd["attention_rnn.bias_ih"] = vec(4 * A); sd["attention_rnn.bias_hh"] = vec(4 * A)
    sd["attention_layer.query_layer.0.weight"] = mat(256, A); sd["attention_layer.query_layer.0.bias"] = vec(256)
    sd["attention_layer.query_layer.2.weight"] = mat(3 * M, 256, 0.5)
    b = vec(3 * M)
    b[M:2 * M] = 1.0
    b[2 * M:] = -0.432
    sd["attention_layer.query_layer.2.bias"] = b
    for i in range(hp["num_decoder_rnn_layer"]):
        kin = E + A if i == 0 else D
        sd[f"decoder_rnn_layers.{i}.weight_ih"] = mat(4 * D, kin); sd[f"decoder_rnn_layers.{i}.weight_hh"] = mat(4 * D, D)
        sd[f"decoder_rnn_layers.{i}.bias_ih"] = vec(4 * D); sd[f"decoder_rnn_layers.{i}.bias_hh"] = vec(4 * D)
    kout = D + E if hp["concat_context_to_last"] else D
    sd["linear_projection.linear_layer.weight"] = mat(nm * r, kout, 2.0); sd["linear_projection.linear_layer.bias"] = vec(nm * r, 0.2)
    sd["stop_layer.linear_layer.weight"] = mat(1, kout, 2.0)
    sd["stop_layer.linear_layer.bias"] = torch.tensor([stop_bias], dtype=torch.float32)
    return sd


PPG2MEL_NET_HP = dict(bnf_dim=144, spk_dim=256, enc_dim=256, downsample_rates=(2, 2), num_mels=80)


def ppg2mel_model_state(hp=PPG2MEL_HP, net_hp=PPG2MEL_NET_HP, seed=0, stop_bias=-2.0):
    """state_dict of models/ppg2mel/__init__.py:MelDecoderMOLv2 (decoder.* from ppg2mel_decoder_state, plus the
    bnf_prenet / pitch_convs / reduce_proj / postnet entries as the reference registers them); BatchNorm running
    statistics away from (0, 1) so the eval-mode affine is exercised."""
    rng = np.random.default_rng(seed + 77)
    E, S, nm = net_hp["enc_dim"], net_hp["spk_dim"], net_hp["num_mels"]
    sd = {"decoder." + k: v for k, v in ppg2mel_decoder_state(hp, seed, stop_bias).items()}

    def t(*shape, fan):
        return torch.from_numpy((rng.standard_normal(shape) / np.sqrt(fan)).astype(np.float32))

    for name, cin in (("bnf_prenet", net_hp["bnf_dim"]), ("pitch_convs", 2)):
        sd[f"{name}.0.weight"] = t(E, cin, 1, fan=cin)
        for idx, r in zip((3, 6), net_hp["downsample_rates"]):
            sd[f"{name}.{idx}.weight"] = t(E, E, 2 * r, fan=E * 2 * r)
            sd[f"{name}.{idx}.bias"] = t(E, fan=100)
    sd["reduce_proj.weight"] = t(E, E + S, fan=E)
    sd["reduce_proj.bias"] = t(E, fan=100)
    dims = [nm, 512, 512, 512, 512, nm]
    for i in range(5):
        p = f"postnet.convolutions.{i}"
        sd[p + ".0.conv.weight"] = t(dims[i + 1], dims[i], 5, fan=dims[i] * 5)
        sd[p + ".0.conv.bias"] = t(dims[i + 1], fan=100)
        sd[p + ".1.weight"] = torch.from_numpy(rng.uniform(0.5, 1.5, dims[i + 1]).astype(np.float32))
        sd[p + ".1.bias"] = t(dims[i + 1], fan=100)
        sd[p + ".1.running_mean"] = t(dims[i + 1], fan=100)
        sd[p + ".1.running_var"] = torch.from_numpy(rng.uniform(0.5, 1.5, dims[i + 1]).astype(np.float32))
        sd[p + ".1.num_batches_tracked"] = torch.tensor(100)
    return sd


def ppg2mel_inputs(batch, t, seed=0, net_hp=PPG2MEL_NET_HP):
    """(bottle_neck_features [B, T, bnf_dim], logf0_uv [B, T, 2], spembs [B, spk_dim]) like the convertor feeds
    MelDecoderMOLv2.inference: PPG posteriors-ish features, normalised log-f0 + a 0/1 voicing flag, a d-vector."""
    rng = np.random.default_rng(seed + 5)
    bnf = rng.standard_normal((batch, t, net_hp["bnf_dim"])).astype(np.float32)
    lf0 = rng.standard_normal((batch, t, 1)).astype(np.float32)
    uv = (rng.uniform(size=(batch, t, 1)) > 0.3).astype(np.float32)
    spk = np.abs(rng.standard_normal((batch, net_hp["spk_dim"]))).astype(np.float32)
    return bnf, np.concatenate([lf0 * uv, uv], axis=-1), spk


PPG2MEL_MODEL_CASES = [  # (name, batch, frames, weight seed, stop bias, input seed, torch RNG seed)
    ("model_b1_t101", 1, 101, 5, -2.0, 1, 11),
    ("model_b2_t96", 2, 96, 3, 0.0, 2, 12),
]


def ppg2mel_memory(batch, t_enc, seed=0, enc_dim=256):
    """Decoder memory [B, T_enc, enc_dim] (the reference feeds InstanceNorm'd features: O(1))."""
    return np.random.default_rng(seed).standard_normal((batch, t_enc, enc_dim)).astype(np.float32)


def ppg2mel_dropout_masks(seed, n_steps, batch, dims=(256, 128)):
    """Bernoulli(0.5) keep masks of the two prenet layers per decoder step, program order (layer 0, layer 1)."""
    g = torch.Generator().manual_seed(seed)
    return [torch.empty(batch, d).bernoulli_(0.5, generator=g) for _ in range(n_steps) for d in dims]


PPG2MEL_CASES = [  # golden cases: (name, batch, t_enc, weight seed, stop bias, memory seed, torch RNG seed)
    ("b1_t30_runs_to_max", 1, 30, 3, -2.0, 1, 7),
    ("b1_t26_stops", 1, 26, 4, 0.0, 2, 8),
    ("b3_t24_batched", 3, 24, 3, 0.0, 3, 7),
]


VITS_DEC = dict(initial_channel=192, resblock="1", resblock_kernel_sizes=[3, 7, 11],
                resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], upsample_rates=[8, 8, 2, 2],
                upsample_initial_channel=512, upsample_kernel_sizes=[16, 16, 4, 4], gin_channels=256)


def vits_dec_state(h=VITS_DEC, seed=0):
    """state_dict of models/synthesizer/models/vits.py:Generator (names as the reference registers them):
    the HiFi-GAN-style stack of gan_state() with a plain conv_pre, a bias-free conv_post and cond."""
    hh = dict(h)
    hh["num_mels"] = h["initial_channel"]
    sd = dict(gan_state(hh, "hifigan", seed=seed)["generator"])
    import oracle.gan as og
    folded = og.fold_weight_norm_state({k: v for k, v in sd.items() if k.startswith("conv_pre.")})
    for k in [k for k in sd if k.startswith("conv_pre.")]:
        del sd[k]
    sd["conv_pre.weight"], sd["conv_pre.bias"] = folded["conv_pre.weight"], folded["conv_pre.bias"]
    post = og.fold_weight_norm_state({k: v for k, v in sd.items() if k.startswith("conv_post.")})
    for k in [k for k in sd if k.startswith("conv_post.")]:
        del sd[k]
    sd["conv_post.weight"] = post["conv_post.weight"]
    rng = np.random.default_rng(seed + 1000)
    gin, c0 = h["gin_channels"], h["upsample_initial_channel"]
    if gin:
        sd["cond.weight"] = torch.from_numpy((rng.standard_normal((c0, gin, 1)) / np.sqrt(gin)).astype(np.float32))
        sd["cond.bias"] = torch.from_numpy((0.05 * rng.standard_normal(c0)).astype(np.float32))
    return sd


def vits_latent(frames, batch=1, seed=0, channels=192, gin=256):
    rng = np.random.default_rng(seed)
    z = rng.standard_normal((batch, channels, frames)).astype(np.float32)
    g = rng.standard_normal((batch, gin, 1)).astype(np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    return z, g


VITS_CASES = [("uic64_t9_b2_g", 64, 9, 2, True, 3), ("uic512_t6_b1_nog", 512, 6, 1, False, 4)]
VITS_DEC_RB2 = dict(VITS_DEC, resblock="2", resblock_kernel_sizes=[3, 5, 7], resblock_dilation_sizes=[[1, 2], [2, 6], [3, 12]])
VITS_RB2_CASES = [("rb2_uic64_t9_b2_g", 64, 9, 2, True, 5), ("rb2_uic256_t6_b1_nog", 256, 6, 1, False, 6)]


# ---------------------------------------------------------------------------- waveform wire format
WAVE_CASES = (("f32", 5000, 1.3, 21), ("f64", 4097, 1.1, 22), ("f32", 777, 0.004, 23), ("f64", 1, 0.7, 24))  # dtype, n, peak, seed


def wave_input(dtype, n, peak, seed):
    """A waveform with clipping excursions (peak > 1), exact +-1, and exact half-way points k + 0.5 of the
    x*32768 grid (round-half-even vs truncation vs round-half-up all differ there)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    x = peak * np.sin(0.031 * t + 0.5) * (0.6 + 0.4 * np.sin(0.0017 * t)) + 0.015 * peak * rng.standard_normal(n)
    if n > 64 and peak >= 0.5:  # (the quiet case keeps max|x| < 0.01: save_wav's floor)
        x[3], x[4] = 1.0, -1.0
        ks = np.array([0, 1, 2, 3, -1, -2, -3, 100, 101, -100, -101, 32766, -32767])
        x[10:10 + len(ks)] = (ks + 0.5) / 32768.0
        x[30] = 32767.0 / 32768.0
        x[31] = 0.0
    return x.astype(np.float32 if dtype == "f32" else np.float64)
