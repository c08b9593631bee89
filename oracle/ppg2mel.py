"""Oracle (test infrastructure only): ppg2mel MoL-attention RNN decoder, inference path.

Restates models/ppg2mel/rnn_decoder_mol.py -- Decoder.inference :267-316 and
Decoder.inference_batched :318-374 (loop body: DecoderPrenet :10-22, attend :187-198, decode :200-209,
projection / stop :287-302) -- and models/ppg2mel/utils/mol_attention.py (MOLAttention.forward :67-122)
as functional fp32 ATen CPU ops on a state_dict `w` of the reference's Decoder module, in the
reference's op order.

Dropout: DecoderPrenet applies F.dropout(p=0.5, training=True) at inference (:20-21); pass masks=None
to draw from the global torch RNG exactly like the reference, or a MaskSource (oracle/tacotron.py) to
inject pre-drawn keep masks.  MOLAttention's own dropout (:90) is off in eval mode.

The one-shot networks around the loop -- MelDecoderMOLv2's bnf_prenet / pitch_convs / reduce_proj
(models/ppg2mel/__init__.py:50-100,166-179) and the CNN postnet (models/ppg2mel/utils/cnn_postnet.py:8-52) --
are restated by encode() / postnet() / model_inference() on the MelDecoderMOLv2 state_dict.

Pinned by tests/golden/ppg2mel.npz (outputs of the reference modules themselves, tests/golden/make_golden.py)."""
import torch
import torch.nn.functional as F

from .tacotron import MaskSource, _dropout  # noqa: F401  (same injected-mask convention)

HP = dict(enc_dim=256, num_mels=80, frames_per_step=2, attention_rnn_dim=512, decoder_rnn_dim=512,
          prenet_dims=(256, 128), num_mixtures=5, encoder_down_factor=4, num_decoder_rnn_layer=1,
          concat_context_to_last=True, eps=1e-5)


def prenet(w, x, masks=None):  # DecoderPrenet.forward :19-22 (bias-free linears)
    i = 0
    while f"prenet.layers.{i}.linear_layer.weight" in w:
        x = _dropout(F.relu(F.linear(x, w[f"prenet.layers.{i}.linear_layer.weight"])), 0.5, masks)
        i += 1
    return x


def lstm_cell(w, name, x, h, c):  # nn.LSTMCell, gate order (i, f, g, o)
    g = F.linear(x, w[name + ".weight_ih"], w[name + ".bias_ih"]) + F.linear(h, w[name + ".weight_hh"], w[name + ".bias_hh"])
    i, f, gg, o = g.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def mol_attention(w, hp, att_h, memory, mu_prev):  # MOLAttention.forward :67-122 (eval: no dropout, no mask)
    M = hp["num_mixtures"]
    q = F.relu(F.linear(att_h, w["attention_layer.query_layer.0.weight"], w["attention_layer.query_layer.0.bias"]))
    mp = F.linear(q, w["attention_layer.query_layer.2.weight"], w["attention_layer.query_layer.2.bias"])
    w_hat, sigma_hat, delta_hat = mp[:, :M], mp[:, M:2 * M], mp[:, 2 * M:3 * M]
    wm = torch.softmax(w_hat, dim=-1) + hp["eps"]
    sigma = F.softplus(sigma_hat) + hp["eps"]
    delta = F.softplus(delta_hat)
    mu_cur = mu_prev + delta
    T = memory.size(1)
    j = (torch.arange(0, T + 2.0) + 0.5)[:T + 1]  # init_states :64, forward :98
    phi = wm.unsqueeze(-1) * (1 / (1 + torch.sigmoid((mu_cur.unsqueeze(-1) - j) / sigma.unsqueeze(-1))))
    alpha = torch.sum(phi, dim=1)
    alpha = alpha[:, 1:] - alpha[:, :-1]
    alpha[alpha == 0] = hp["eps"]
    context = torch.bmm(alpha.unsqueeze(1), memory).squeeze(1)
    return context, alpha, mu_cur


def inference_batched(w, hp, memory, stop_threshold=0.5, masks=None, max_steps=None):
    """Loop of Decoder.inference_batched :318-360 (B >= 1).  Returns the per-step tensors
    (mel [B, steps*r, num_mels], alignments [B, steps, T_enc], stop logits [B, steps]) BEFORE the
    reference's per-item truncation at the first frame above the threshold (:364-372)."""
    B, T, _ = memory.shape
    A, Dd, nm, r = hp["attention_rnn_dim"], hp["decoder_rnn_dim"], hp["num_mels"], hp["frames_per_step"]
    L = hp["num_decoder_rnn_layer"]
    att_h, att_c = torch.zeros(B, A), torch.zeros(B, A)
    dec_h = [torch.zeros(B, Dd) for _ in range(L)]
    dec_c = [torch.zeros(B, Dd) for _ in range(L)]
    ctx = torch.zeros(B, hp["enc_dim"])
    mu = torch.zeros(B, hp["num_mixtures"])
    x = torch.zeros(B, nm)  # go frame :111-115
    max_step = T * hp["encoder_down_factor"] // r if max_steps is None else max_steps
    min_step = T * hp["encoder_down_factor"] // r - 5
    mels, aligns, stops = [], [], []
    while True:
        p = prenet(w, x, masks)
        att_h, att_c = lstm_cell(w, "attention_rnn", torch.cat((p, ctx), -1), att_h, att_c)  # attend :188-190
        ctx, alpha, mu = mol_attention(w, hp, att_h, memory, mu)
        d_in = torch.cat((att_h, ctx), -1)
        for i in range(L):  # decode :200-209
            dec_h[i], dec_c[i] = lstm_cell(w, f"decoder_rnn_layers.{i}", d_in if i == 0 else dec_h[i - 1], dec_h[i], dec_c[i])
        out = dec_h[-1]
        if hp["concat_context_to_last"]:
            out = torch.cat((out, ctx), dim=1)
        mel = F.linear(out, w["linear_projection.linear_layer.weight"], w["linear_projection.linear_layer.bias"])
        stop = F.linear(out, w["stop_layer.linear_layer.weight"], w["stop_layer.linear_layer.bias"])
        mels.append(mel)
        aligns.append(alpha)
        stops.append(stop[:, 0])
        if bool(torch.all(torch.sigmoid(stop[:, 0]) > stop_threshold)) and len(mels) >= min_step:
            break
        if len(mels) >= max_step:
            break
        x = mel[:, -nm:]
    mel_out = torch.stack(mels).transpose(0, 1).contiguous().view(B, -1, nm)
    return mel_out, torch.stack(aligns).transpose(0, 1), torch.stack(stops).transpose(0, 1)


def inference(w, hp, memory, stop_threshold=0.5, masks=None):
    """Decoder.inference :267-316 (memory [1, T_enc, enc_dim]) -> (mel [1, steps*r, num_mels], alignments [1, steps, T_enc])."""
    mel, al, _ = inference_batched(w, hp, memory, stop_threshold, masks)
    return mel, al


def inference_batched_cut(w, hp, memory, stop_threshold=0.5, masks=None):
    """Decoder.inference_batched :318-374 including the per-item cut at the first step above the threshold
    (:369-373; IndexError, like the reference, for an utterance that never crosses it)."""
    mel, al, stop = inference_batched(w, hp, memory, stop_threshold, masks)
    parts = []
    for b in range(mel.size(0)):
        idx = int(torch.nonzero(torch.sigmoid(stop[b]) > stop_threshold)[0][0])
        parts.append(mel[b, :idx, :])
    return torch.cat(parts, dim=0).unsqueeze(0), al


NET_HP = dict(bnf_dim=144, spk_dim=256, enc_dim=256, downsample_rates=(2, 2), num_mels=80)


def instance_norm(x, eps=1e-5):  # nn.InstanceNorm1d(affine=False): per (utterance, channel) row, biased variance
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


def _down_branch(w, name, x, rates):  # bnf_prenet / pitch_convs Sequential, models/ppg2mel/__init__.py:50-98
    x = instance_norm(F.leaky_relu(F.conv1d(x, w[name + ".0.weight"]), 0.1))
    for idx, r in zip((3, 6), rates):
        x = F.conv1d(x, w[f"{name}.{idx}.weight"], w[f"{name}.{idx}.bias"], stride=r, padding=r // 2)
        x = instance_norm(F.leaky_relu(x, 0.1))
    return x


def encode(w, hp, bnf, logf0_uv, spembs):
    """MelDecoderMOLv2.inference :172-179: bnf [B, T, bnf_dim], logf0_uv [B, T, 2], spembs [B, spk_dim]
    -> decoder memory [B, T_enc, enc_dim]."""
    rates = hp["downsample_rates"]
    x = _down_branch(w, "bnf_prenet", bnf.transpose(1, 2), rates).transpose(1, 2)
    x = x + _down_branch(w, "pitch_convs", logf0_uv.transpose(1, 2), rates).transpose(1, 2)
    spk = F.normalize(spembs).unsqueeze(1).expand(-1, x.size(1), -1)
    return F.linear(torch.cat([x, spk], dim=-1), w["reduce_proj.weight"], w["reduce_proj.bias"])


def postnet(w, mel):
    """mel_outputs + Postnet(mel_outputs) (:186-187; cnn_postnet.py:47-52, eval: BatchNorm running stats, no dropout).
    mel [B, T, num_mels] -> same shape."""
    x = mel.transpose(1, 2)
    n = 0
    while f"postnet.convolutions.{n}.0.conv.weight" in w:
        n += 1
    for i in range(n):
        p = f"postnet.convolutions.{i}"
        k = w[p + ".0.conv.weight"].shape[-1]
        x = F.conv1d(x, w[p + ".0.conv.weight"], w[p + ".0.conv.bias"], padding=(k - 1) // 2)
        x = F.batch_norm(x, w[p + ".1.running_mean"], w[p + ".1.running_var"], w[p + ".1.weight"], w[p + ".1.bias"],
                         False, 0.0, 1e-5)
        if i < n - 1:
            x = torch.tanh(x)
    return mel + x.transpose(1, 2)


def model_inference(w, hp, net_hp, bnf, logf0_uv, spembs, masks=None):
    """MelDecoderMOLv2.inference :166-192 -> (mel_outputs[0], mel_outputs_postnet[0], alignments[0])."""
    memory = encode(w, net_hp, bnf, logf0_uv, spembs)
    dw = {k[len("decoder."):]: v for k, v in w.items() if k.startswith("decoder.")}
    if memory.size(0) > 1:
        mel, al = inference_batched_cut(dw, hp, memory, masks=masks)
    else:
        mel, al = inference(dw, hp, memory, masks=masks)
    return mel[0], postnet(w, mel)[0], al[0]
