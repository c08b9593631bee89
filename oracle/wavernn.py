"""Oracle: WaveRNN (fatchord) generate() -- fp32 ATen CPU kernels + numpy float64 post.

`w` maps the reference module's state_dict names to tensors
(models/vocoder/wavernn/models/fatchord_version.py:88-122).  `hp` is a dict
with the keys of models/vocoder/wavernn/hparams.py that generate() reads.
"""
import numpy as np
import torch
import torch.nn.functional as F
from scipy.signal import lfilter

HP = dict(rnn_dims=512, fc_dims=512, bits=9, pad=2, upsample_factors=(5, 5, 8), feat_dims=80,
          compute_dims=128, res_out_dims=128, res_blocks=10, hop_length=256, sample_rate=16000,
          mode="RAW", mu_law=True, apply_preemphasis=True, preemphasis=0.97, mel_max_abs_value=4.0)


def _bn(w, p, x):  # nn.BatchNorm1d in eval mode
    return F.batch_norm(x, w[p + ".running_mean"], w[p + ".running_var"], w[p + ".weight"],
                        w[p + ".bias"], False, 0.0, 1e-5)


def mel_resnet(w, hp, x):
    """MelResNet.forward fatchord_version.py:37-44 (ResBlock :17-24)."""
    p = "upsample.resnet."
    x = F.conv1d(x, w[p + "conv_in.weight"])
    x = F.relu(_bn(w, p + "batch_norm", x))
    for i in range(hp["res_blocks"]):
        q = f"{p}layers.{i}."
        r = x
        x = F.relu(_bn(w, q + "batch_norm1", F.conv1d(x, w[q + "conv1.weight"])))
        x = _bn(w, q + "batch_norm2", F.conv1d(x, w[q + "conv2.weight"]))
        x = x + r
    return F.conv1d(x, w[p + "conv_out.weight"], w[p + "conv_out.bias"])


def _stretch(x, x_scale):  # Stretch2d(x_scale, 1) :47-57 on [b, c, h, w]
    b, c, h, ww = x.size()
    x = x.unsqueeze(-1).unsqueeze(3).repeat(1, 1, 1, 1, 1, x_scale)
    return x.view(b, c, h, ww * x_scale)


def upsample_network(w, hp, m):
    """UpsampleNetwork.forward :78-85.  m [1, 80, F+2*pad] -> (mels [1, L, 80], aux [1, L, 128])."""
    total_scale = int(np.prod(hp["upsample_factors"]))
    indent = hp["pad"] * total_scale
    aux = mel_resnet(w, hp, m).unsqueeze(1)
    aux = _stretch(aux, total_scale).squeeze(1)
    m = m.unsqueeze(1)
    for i, scale in enumerate(hp["upsample_factors"]):
        m = _stretch(m, scale)
        m = F.conv2d(m, w[f"upsample.up_layers.{2 * i + 1}.weight"], padding=(0, scale))
    m = m.squeeze(1)[:, :, indent:-indent]
    return m.transpose(1, 2), aux.transpose(1, 2)


def pad_tensor(x, pad, side="both"):  # :273-286
    b, t, c = x.size()
    total = t + 2 * pad if side == "both" else t + pad
    padded = torch.zeros(b, total, c)
    if side in ("before", "both"):
        padded[:, pad:pad + t, :] = x
    elif side == "after":
        padded[:, :t, :] = x
    return padded


def fold_with_overlap(x, target, overlap):  # :288-338
    _, total_len, features = x.size()
    num_folds = (total_len - overlap) // (target + overlap)
    extended_len = num_folds * (overlap + target) + overlap
    remaining = total_len - extended_len
    if remaining != 0:
        num_folds += 1
        padding = target + 2 * overlap - remaining
        x = pad_tensor(x, padding, side="after")
    folded = torch.zeros(num_folds, target + 2 * overlap, features)
    for i in range(num_folds):
        start = i * (target + overlap)
        end = start + target + 2 * overlap
        folded[i] = x[:, start:end, :]
    return folded


def xfade_and_unfold(y, target, overlap):  # :340-402
    num_folds, length = y.shape
    target = length - 2 * overlap
    total_len = num_folds * (target + overlap) + overlap
    silence_len = overlap // 2
    fade_len = overlap - silence_len
    silence = np.zeros((silence_len), dtype=np.float64)
    t = np.linspace(-1, 1, fade_len, dtype=np.float64)
    fade_in = np.sqrt(0.5 * (1 + t))
    fade_out = np.sqrt(0.5 * (1 - t))
    fade_in = np.concatenate([silence, fade_in])
    fade_out = np.concatenate([fade_out, silence])
    y[:, :overlap] *= fade_in
    y[:, -overlap:] *= fade_out
    unfolded = np.zeros((total_len), dtype=np.float64)
    for i in range(num_folds):
        start = i * (target + overlap)
        end = start + target + 2 * overlap
        unfolded[start:end] += y[i]
    return unfolded


def decode_mu_law(y, mu):  # wavernn/audio.py:102-107 with from_labels=False
    mu = mu - 1
    return np.sign(y) / mu * ((1 + mu) ** np.abs(y) - 1)


def de_emphasis(x, coef):  # wavernn/audio.py:92-93
    return lfilter([1], [1, -coef], x)


def _gru_cell(w, name, x, h):  # get_gru_cell :265-271 -> nn.GRUCell
    return torch.gru_cell(x, h, w[name + ".weight_ih_l0"], w[name + ".weight_hh_l0"],
                          w[name + ".bias_ih_l0"], w[name + ".bias_hh_l0"])


LOG_SCALE_MIN = float(np.log(1e-14))  # distribution.py:96-97


def sample_mol(logits, u=None):
    """sample_from_discretized_mix_logistic (models/vocoder/distribution.py:87-123) as WaveRNN.generate calls it
    (fatchord_version.py:213-214: y = logits.unsqueeze(0).transpose(1, 2), i.e. B = 1, C = 3 * nr_mix, T = folds).
    logits [b, 3 * nr_mix] -> sample [b] in [-1, 1].
    u: None -> the two uniform_(1e-5, 1 - 1e-5) draws on the global torch RNG in the reference's order
       (mixture indicator [1, b, nr_mix] first, then the logistic [1, b]); else [b, nr_mix + 1] pre-drawn
       uniforms: columns 0..nr_mix-1 the indicator draws, column nr_mix the logistic draw."""
    y = logits.unsqueeze(0).transpose(1, 2)  # B x C x T
    nr_mix = y.size(1) // 3
    y = y.transpose(1, 2)  # B x T x C
    logit_probs = y[:, :, :nr_mix]
    if u is None:
        temp = logit_probs.data.new(logit_probs.size()).uniform_(1e-5, 1.0 - 1e-5)
    else:
        temp = u[:, :nr_mix].unsqueeze(0)
    temp = logit_probs.data - torch.log(-torch.log(temp))
    _, argmax = temp.max(dim=-1)
    one_hot = torch.zeros(argmax.size() + (nr_mix,)).scatter_(len(argmax.size()), argmax.unsqueeze(-1), 1.0)
    means = torch.sum(y[:, :, nr_mix:2 * nr_mix] * one_hot, dim=-1)
    log_scales = torch.clamp(torch.sum(y[:, :, 2 * nr_mix:3 * nr_mix] * one_hot, dim=-1), min=LOG_SCALE_MIN)
    if u is None:
        uu = means.data.new(means.size()).uniform_(1e-5, 1.0 - 1e-5)
    else:
        uu = u[:, nr_mix].unsqueeze(0)
    x = means + torch.exp(log_scales) * (torch.log(uu) - torch.log(1. - uu))
    x = torch.clamp(torch.clamp(x, min=-1.), max=1.)
    return x.view(-1)


def n_classes_of(hp):  # fatchord_version.py:95-98
    return 30 if hp["mode"] == "MOL" else 2 ** hp["bits"]


def sample_loop(w, hp, mels, aux, noise=None, forced=None, return_logits=False, max_steps=None, state=None, return_state=False):
    """Loop body :176-234.  mels [b, T, 80], aux [b, T, 128].
    state / return_state (test infrastructure, no counterpart in the reference): (h1, h2, x) carried from a previous
    call over the steps before mels[:, 0] -- a long replay runs as consecutive windows with bounded memory; state=None
    is the reference's zero initialisation (:179-181).
    RAW mode -- noise: None -> Categorical(p).sample() on the global torch RNG (the reference's call);
           else [T, b, C] Exp(1) draws, sample = argmax(p / noise) (the same arithmetic
           torch.multinomial(p, 1) performs on CPU).
    MOL mode (:213-220) -- noise: None -> global RNG, else [T, b, nr_mix + 1] uniforms (sample_mol).
    forced: optional [b, T] samples fed back instead of the drawn ones (teacher forcing)."""
    n_classes = n_classes_of(hp)
    b_size, seq_len, _ = mels.size()
    if max_steps is not None:
        seq_len = min(seq_len, max_steps)
    if state is None:
        h1 = torch.zeros(b_size, hp["rnn_dims"])
        h2 = torch.zeros(b_size, hp["rnn_dims"])
        x = torch.zeros(b_size, 1)
    else:
        h1, h2, x = state
    d = hp["res_out_dims"] // 4
    aux_split = [aux[:, :, d * i:d * (i + 1)] for i in range(4)]
    output, logits_all = [], []
    for i in range(seq_len):
        m_t = mels[:, i, :]
        a1_t, a2_t, a3_t, a4_t = (a[:, i, :] for a in aux_split)
        x = torch.cat([x, m_t, a1_t], dim=1)
        x = F.linear(x, w["I.weight"], w["I.bias"])
        h1 = _gru_cell(w, "rnn1", x, h1)
        x = x + h1
        inp = torch.cat([x, a2_t], dim=1)
        h2 = _gru_cell(w, "rnn2", inp, h2)
        x = x + h2
        x = torch.cat([x, a3_t], dim=1)
        x = F.relu(F.linear(x, w["fc1.weight"], w["fc1.bias"]))
        x = torch.cat([x, a4_t], dim=1)
        x = F.relu(F.linear(x, w["fc2.weight"], w["fc2.bias"]))
        logits = F.linear(x, w["fc3.weight"], w["fc3.bias"])
        if return_logits:
            logits_all.append(logits)
        if hp["mode"] == "MOL":
            sample = sample_mol(logits, None if noise is None else noise[i])
        else:
            posterior = F.softmax(logits, dim=1)
            if noise is None:
                k = torch.distributions.Categorical(posterior).sample()
            else:
                k = (posterior / noise[i]).argmax(dim=1)
            sample = 2 * k.float() / (n_classes - 1.) - 1.
        output.append(sample)
        x = (forced[:, i] if forced is not None else sample).unsqueeze(-1)
    out = torch.stack(output).transpose(0, 1)
    res = (out, torch.stack(logits_all)) if return_logits else out
    if return_state:
        return (*res, (h1, h2, x)) if return_logits else (res, (h1, h2, x))
    return res


def conditioning(w, hp, mel, batched, target, overlap):
    """generate() :165-174: pad, upsample, fold.  mel [1, 80, F]."""
    mels = pad_tensor(mel.transpose(1, 2), pad=hp["pad"], side="both")
    mels, aux = upsample_network(w, hp, mels.transpose(1, 2))
    if batched:
        mels = fold_with_overlap(mels, target, overlap)
        aux = fold_with_overlap(aux, target, overlap)
    return mels, aux


def postprocess(hp, output, wave_len, batched, target, overlap):
    """generate() :236-257 from the stacked samples [b, T] (torch) to the float64 waveform."""
    output = output.cpu().numpy().astype(np.float64)
    output = xfade_and_unfold(output, target, overlap) if batched else output[0]
    if hp["mu_law"] and hp["mode"] == "RAW":  # :154: mu_law = mu_law if self.mode == 'RAW' else False
        output = decode_mu_law(output, n_classes_of(hp))
    if hp["apply_preemphasis"]:
        output = de_emphasis(output, hp["preemphasis"])
    fade_out = np.linspace(1, 0, 20 * hp["hop_length"])
    output = output[:wave_len]
    output[-20 * hp["hop_length"]:] *= fade_out
    return output


def consume_gru_init_rng(hp):
    """generate() re-wraps both GRUs as fresh nn.GRUCell objects on EVERY call (:160-161,
    :265-271); their constructors draw the default uniform init from the global torch RNG
    before the weights are overwritten.  Anything that wants the reference's sample stream
    for a given torch.manual_seed must consume the same draws first."""
    d = hp["res_out_dims"] // 4
    torch.nn.GRUCell(hp["rnn_dims"], hp["rnn_dims"])
    torch.nn.GRUCell(hp["rnn_dims"] + d, hp["rnn_dims"])


def generate(w, hp, mel, batched, target, overlap, noise=None):
    """WaveRNN.generate :153-257.  mel [1, 80, F] float32 (already normalised).
    With noise=None the global torch RNG is consumed exactly like the reference does."""
    if noise is None:
        consume_gru_init_rng(hp)
    with torch.no_grad():
        wave_len = (mel.size(-1) - 1) * hp["hop_length"]
        mels, aux = conditioning(w, hp, mel, batched, target, overlap)
        out = sample_loop(w, hp, mels, aux, noise)
    return postprocess(hp, out, wave_len, batched, target, overlap)


def infer_waveform(w, hp, mel, normalize=True, batched=True, target=8000, overlap=800, noise=None):
    """models/vocoder/wavernn/inference.py:45-64."""
    if normalize:
        mel = mel / hp["mel_max_abs_value"]
    mel = torch.from_numpy(mel[None, ...])
    return generate(w, hp, mel, batched, target, overlap, noise), hp["sample_rate"]
