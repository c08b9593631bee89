"""Oracle (test infrastructure only): ppg2mel MoL-attention RNN decoder, inference path.

Restates models/ppg2mel/rnn_decoder_mol.py -- Decoder.inference :267-316 and
Decoder.inference_batched :318-374 (loop body: DecoderPrenet :10-22, attend :187-198, decode :200-209,
projection / stop :287-302) -- and models/ppg2mel/utils/mol_attention.py (MOLAttention.forward :67-122)
as functional fp32 ATen CPU ops on a state_dict `w` of the reference's Decoder module, in the
reference's op order.

Dropout: DecoderPrenet applies F.dropout(p=0.5, training=True) at inference (:20-21); pass masks=None
to draw from the global torch RNG exactly like the reference, or a MaskSource (oracle/tacotron.py) to
inject pre-drawn keep masks.  MOLAttention's own dropout (:90) is off in eval mode.

Pinned by tests/golden/ppg2mel.npz (outputs of the reference module itself, tests/golden/make_golden.py)."""
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
