"""Oracle: SV2TTS Tacotron inference -- Tacotron.generate() and the Synthesizer facade glue.

fp32 ATen CPU kernels in the reference's op order.  `w` maps the reference module's
state_dict names (models/synthesizer/models/tacotron.py) to tensors.  Dropout: the reference's
PreNet applies F.dropout(training=True) at inference (sublayer/pre_net.py:23,26); pass
`masks=None` to draw from the global torch RNG exactly like the reference, or a MaskSource to
inject pre-drawn keep masks (x * mask * 2 == F.dropout(x, 0.5, True) for the same draws,
SURVEY.md section 8c)."""
import numpy as np
import torch
import torch.nn.functional as F

HP = dict(num_chars=75, embed_dims=512, encoder_dims=256, decoder_dims=128, n_mels=80, postnet_dims=512,
          encoder_K=5, lstm_dims=1024, postnet_K=5, num_highways=4, dropout=0.5, speaker_embedding_size=256,
          gst_E=512, gst_heads=8, gst_tokens=10, max_r=20, stop_threshold=-3.4, synthesis_batch_size=16,
          lsa_kernel=31, lsa_filters=32)


class MaskSource:
    """Pre-drawn Bernoulli(0.5) keep masks consumed in program order."""

    def __init__(self, masks):
        self.masks, self.i = list(masks), 0

    def next(self, shape):
        m = self.masks[self.i]
        self.i += 1
        assert tuple(m.shape) == tuple(shape), (m.shape, shape)
        return m


def _dropout(x, p, masks):
    if masks is None:
        return F.dropout(x, p, training=True)
    return x * masks.next(x.shape) * (1.0 / (1.0 - p))


def prenet(w, p, x, drop, masks=None):  # sublayer/pre_net.py:12-27
    x = F.relu(F.linear(x, w[p + ".fc1.weight"], w[p + ".fc1.bias"]))
    x = _dropout(x, drop, masks)
    x = F.relu(F.linear(x, w[p + ".fc2.weight"], w[p + ".fc2.bias"]))
    return _dropout(x, drop, masks)


def _bnconv(w, p, x, k, relu=True):  # common/batch_norm_conv.py:11-14 (BN AFTER ReLU)
    x = F.conv1d(x, w[p + ".conv.weight"], None, padding=k // 2)
    x = F.relu(x) if relu else x
    return F.batch_norm(x, w[p + ".bnorm.running_mean"], w[p + ".bnorm.running_var"], w[p + ".bnorm.weight"],
                        w[p + ".bnorm.bias"], False, 0.0, 1e-5)


def _gru_seq(w, p, x, reverse=False):
    """One direction of nn.GRU(batch_first) from h0 = 0.  x [B, T, C] -> [B, T, H]."""
    sfx = "_reverse" if reverse else ""
    wi, wh = w[f"{p}.weight_ih_l0{sfx}"], w[f"{p}.weight_hh_l0{sfx}"]
    bi, bh = w[f"{p}.bias_ih_l0{sfx}"], w[f"{p}.bias_hh_l0{sfx}"]
    B, T, _ = x.shape
    h = torch.zeros(B, wh.shape[1])
    out = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        h = torch.gru_cell(x[:, t], h, wi, wh, bi, bh)
        out[t] = h
    return torch.stack(out, dim=1)


def cbhg(w, p, x, K, num_highways, use_torch_gru=True):
    """CBHG.forward sublayer/cbhg.py:40-78.  x [B, C_in, T] -> [B, T, channels]."""
    residual = x
    seq_len = x.size(-1)
    bank = [_bnconv(w, f"{p}.conv1d_bank.{k - 1}", x, k)[:, :, :seq_len] for k in range(1, K + 1)]
    x = torch.cat(bank, dim=1)
    x = F.max_pool1d(x, kernel_size=2, stride=1, padding=1)[:, :, :seq_len]
    x = _bnconv(w, p + ".conv_project1", x, 3)
    x = _bnconv(w, p + ".conv_project2", x, 3, relu=False)
    x = x + residual
    x = x.transpose(1, 2)
    if p + ".pre_highway.weight" in w:  # highway_mismatch cbhg.py:26-30
        x = F.linear(x, w[p + ".pre_highway.weight"])
    for i in range(num_highways):  # common/highway_network.py:12-17
        q = f"{p}.highways.{i}"
        x1 = F.linear(x, w[q + ".W1.weight"], w[q + ".W1.bias"])
        x2 = F.linear(x, w[q + ".W2.weight"], w[q + ".W2.bias"])
        g = torch.sigmoid(x2)
        x = g * F.relu(x1) + (1. - g) * x
    if use_torch_gru:
        names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
        flat = [w[f"{p}.rnn.{n}"] for n in names] + [w[f"{p}.rnn.{n}_reverse"] for n in names]
        h0 = torch.zeros(2, x.size(0), flat[1].shape[1])
        out, _ = torch.gru(x, h0, flat, True, 1, 0.0, False, True, True)
        return out
    return torch.cat([_gru_seq(w, p + ".rnn", x), _gru_seq(w, p + ".rnn", x, True)], dim=2)


def encoder(w, hp, texts, masks=None):  # tacotron.py:31-44
    x = F.embedding(texts, w["encoder.embedding.weight"])
    x = prenet(w, "encoder.pre_net", x, hp["dropout"], masks)
    x = x.transpose(1, 2)
    return cbhg(w, "encoder.cbhg", x, hp["encoder_K"], hp["num_highways"])


def _mha(w, p, query, key, key_dim, num_heads, num_units):  # global_style_token.py:125-145
    querys = F.linear(query, w[p + ".W_query.weight"])
    keys = F.linear(key, w[p + ".W_key.weight"])
    values = F.linear(key, w[p + ".W_value.weight"])
    split = num_units // num_heads
    querys = torch.stack(torch.split(querys, split, dim=2), dim=0)
    keys = torch.stack(torch.split(keys, split, dim=2), dim=0)
    values = torch.stack(torch.split(values, split, dim=2), dim=0)
    scores = torch.matmul(querys, keys.transpose(2, 3)) / (key_dim ** 0.5)
    scores = F.softmax(scores, dim=3)
    out = torch.matmul(scores, values)
    return torch.cat(torch.split(out, 1, dim=0), dim=3).squeeze(0)


def gst_style_embed(w, hp, speaker_embedding, style_idx):
    """Inference branches of tacotron.py:243-252."""
    E, heads = hp["gst_E"], hp["gst_heads"]
    if 0 <= style_idx < 10:
        query = torch.zeros(1, 1, E)
        key = torch.tanh(w["gst.stl.embed"])[style_idx].unsqueeze(0).expand(1, -1, -1)
        return _mha(w, "gst.stl.attention", query, key, E // heads, heads, E)
    # ReferenceEncoder on zeros + speaker embedding -> STL (global_style_token.py:20-28,58-76,93-103)
    B = speaker_embedding.size(0)
    inputs = torch.zeros(B, 1, hp["speaker_embedding_size"])
    out = inputs.view(B, 1, -1, 256)  # gst_hyperparameters.n_mels = 256
    for i in range(6):
        out = F.conv2d(out, w[f"gst.encoder.convs.{i}.weight"], w[f"gst.encoder.convs.{i}.bias"], stride=2, padding=1)
        out = F.batch_norm(out, w[f"gst.encoder.bns.{i}.running_mean"], w[f"gst.encoder.bns.{i}.running_var"],
                           w[f"gst.encoder.bns.{i}.weight"], w[f"gst.encoder.bns.{i}.bias"], False, 0.0, 1e-5)
        out = F.relu(out)
    out = out.transpose(1, 2)
    T = out.size(1)
    out = out.contiguous().view(B, T, -1)
    h = torch.zeros(B, E // 2)
    for t in range(T):
        h = torch.gru_cell(out[:, t], h, w["gst.encoder.gru.weight_ih_l0"], w["gst.encoder.gru.weight_hh_l0"],
                           w["gst.encoder.gru.bias_ih_l0"], w["gst.encoder.gru.bias_hh_l0"])
    enc_out = torch.cat([h, speaker_embedding], dim=-1)  # use_ser_for_gst
    keys = torch.tanh(w["gst.stl.embed"]).unsqueeze(0).expand(B, -1, -1)
    return _mha(w, "gst.stl.attention", enc_out.unsqueeze(1), keys, E // heads, heads, E)


def add_speaker_embedding(x, speaker_embedding):  # tacotron.py:171-197
    B, T = x.size(0), x.size(1)
    S = speaker_embedding.size(1)
    e = speaker_embedding.repeat_interleave(T, dim=1).reshape(B, S, T).transpose(1, 2)
    return torch.cat((x, e), 2)


def encoder_memory(w, hp, texts, speaker_embedding, style_idx, masks=None):
    """tacotron.py:234-255 -> (encoder_seq [B,T,P], encoder_seq_proj [B,T,128])."""
    enc = encoder(w, hp, texts, masks)
    enc = add_speaker_embedding(enc, speaker_embedding)
    style = gst_style_embed(w, hp, speaker_embedding, style_idx)
    style = style.expand(enc.size(0), enc.size(1), -1)
    enc = torch.cat([enc, style], dim=-1)
    return enc, F.linear(enc, w["encoder_proj.weight"])


class LSAState:
    def __init__(self, b, t):
        self.cumulative = torch.zeros(b, t)
        self.attention = torch.zeros(b, t)


def lsa(w, st, encoder_seq_proj, query, chars):  # sublayer/lsa.py:21-42
    p = "decoder.attn_net"
    pq = F.linear(query, w[p + ".W.weight"], w[p + ".W.bias"]).unsqueeze(1)
    loc = F.conv1d(st.cumulative.unsqueeze(1), w[p + ".conv.weight"], w[p + ".conv.bias"],
                   padding=(w[p + ".conv.weight"].shape[2] - 1) // 2)
    ploc = F.linear(loc.transpose(1, 2), w[p + ".L.weight"])
    u = F.linear(torch.tanh(pq + encoder_seq_proj + ploc), w[p + ".v.weight"]).squeeze(-1)
    u = u * (chars != 0).float()
    scores = F.softmax(u, dim=1)
    st.attention = scores
    st.cumulative = st.cumulative + st.attention
    return scores.unsqueeze(-1).transpose(1, 2)


def decoder_step(w, hp, r, st, encoder_seq, encoder_seq_proj, prenet_in, hidden, cells, context, chars, masks):
    """Decoder.forward tacotron.py:71-138 (eval: no zoneout)."""
    attn_h, h1, h2 = hidden
    c1, c2 = cells
    B = encoder_seq.size(0)
    po = prenet(w, "decoder.prenet", prenet_in, hp["dropout"], masks)
    attn_in = torch.cat([context, po], dim=-1)
    attn_h = torch.gru_cell(attn_in, attn_h, w["decoder.attn_rnn.weight_ih"], w["decoder.attn_rnn.weight_hh"],
                            w["decoder.attn_rnn.bias_ih"], w["decoder.attn_rnn.bias_hh"])
    scores = lsa(w, st, encoder_seq_proj, attn_h, chars)
    context = (scores @ encoder_seq).squeeze(1)
    x = torch.cat([context, attn_h], dim=1)
    x = F.linear(x, w["decoder.rnn_input.weight"], w["decoder.rnn_input.bias"])
    h1, c1 = torch.lstm_cell(x, (h1, c1), w["decoder.res_rnn1.weight_ih"], w["decoder.res_rnn1.weight_hh"],
                             w["decoder.res_rnn1.bias_ih"], w["decoder.res_rnn1.bias_hh"])
    x = x + h1
    h2, c2 = torch.lstm_cell(x, (h2, c2), w["decoder.res_rnn2.weight_ih"], w["decoder.res_rnn2.weight_hh"],
                             w["decoder.res_rnn2.bias_ih"], w["decoder.res_rnn2.bias_hh"])
    x = x + h2
    mels = F.linear(x, w["decoder.mel_proj.weight"]).view(B, hp["n_mels"], hp["max_r"])[:, :, :r]
    s = torch.cat((x, context), dim=1)
    stop = torch.sigmoid(F.linear(s, w["decoder.stop_proj.weight"], w["decoder.stop_proj.bias"]))
    return mels, scores, (attn_h, h1, h2), (c1, c2), context, stop


def decode(w, hp, r, encoder_seq, encoder_seq_proj, chars, steps, min_stop_token, masks=None):
    """Decoder loop tacotron.py:219-278 -> (mel_outputs [B,80,F], attn [B,F/r,T])."""
    B, T, P = encoder_seq.shape
    hidden = (torch.zeros(B, hp["decoder_dims"]), torch.zeros(B, hp["lstm_dims"]), torch.zeros(B, hp["lstm_dims"]))
    cells = (torch.zeros(B, hp["lstm_dims"]), torch.zeros(B, hp["lstm_dims"]))
    go = torch.zeros(B, hp["n_mels"])
    context = torch.zeros(B, P)
    st = LSAState(B, T)
    mel_outputs, attn_scores = [], []
    for t in range(0, steps, r):
        prenet_in = mel_outputs[-1][:, :, -1] if t > 0 else go
        mels, scores, hidden, cells, context, stop = decoder_step(
            w, hp, r, st, encoder_seq, encoder_seq_proj, prenet_in, hidden, cells, context, chars, masks)
        mel_outputs.append(mels)
        attn_scores.append(scores)
        if (stop * 10 > min_stop_token).all() and t > 10:
            break
    return torch.cat(mel_outputs, dim=2), torch.cat(attn_scores, 1)


def postnet(w, hp, mel_outputs):  # tacotron.py:281-283
    x = cbhg(w, "postnet", mel_outputs, hp["postnet_K"], hp["num_highways"])
    return F.linear(x, w["post_proj.weight"]).transpose(1, 2)


def generate(w, hp, r, texts, speaker_embedding, steps=2000, style_idx=0, min_stop_token=5, masks=None):
    """Tacotron.generate tacotron.py:295-298 -> (mel_outputs, linear, attn_scores)."""
    with torch.no_grad():
        enc, enc_proj = encoder_memory(w, hp, texts, speaker_embedding, style_idx, masks)
        mel, attn = decode(w, hp, r, enc, enc_proj, texts, steps, min_stop_token, masks)
        linear = postnet(w, hp, mel)
    return mel, linear, attn


def synthesize_spectrograms(w, hp, r, token_seqs, embeddings, style_idx=0, min_stop_token=5, steps=2000, masks=None):
    """Synthesizer.synthesize_spectrograms models/synthesizer/inference.py:103-142 from token id
    sequences (the pinyin/text front-end stays on the reference CPU path)."""
    bs = hp["synthesis_batch_size"]
    specs, alignments = [], None
    for i in range(0, len(token_seqs), bs):
        batch = token_seqs[i:i + bs]
        max_len = max(len(t) for t in batch)
        chars = np.stack([np.pad(t, (0, max_len - len(t)), mode="constant") for t in batch])
        chars = torch.tensor(chars).long()
        spk = torch.tensor(np.stack(embeddings[i:i + bs])).float()
        _, mels, alignments = generate(w, hp, r, chars, spk, steps, style_idx, min_stop_token, masks)
        for m in mels.numpy():
            while np.max(m[:, -1]) < hp["stop_threshold"]:
                m = m[:, :-1]
            specs.append(m)
    return specs, alignments
