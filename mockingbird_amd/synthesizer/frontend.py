"""Once-per-chunk front half of Tacotron.forward (tacotron.py:234-255): text encoder, speaker
embedding tiling, GST style embedding, encoder_proj.  It runs once per batch, outside the
autoregressive hot loop; SURVEY.md section 8(a) T3/T4 keep it on PyTorch-ROCm tensor ops for the first
pass (section 8(f) rank 1 moves it onto the HIP conv/GRU kernels).  Produces the attention memory the
HIP decoder (mb_taco_decode) consumes."""
import torch
import torch.nn.functional as F


def _dropout(x, p, masks):
    if masks is None:
        return F.dropout(x, p, training=True)  # always on, pre_net.py:23,26
    m = masks.pop(0).to(x.device)
    return x * m * (1.0 / (1.0 - p))


def _prenet(w, p, x, drop, masks):
    x = _dropout(F.relu(F.linear(x, w[p + ".fc1.weight"], w[p + ".fc1.bias"])), drop, masks)
    return _dropout(F.relu(F.linear(x, w[p + ".fc2.weight"], w[p + ".fc2.bias"])), drop, masks)


def _bnconv(w, p, x, k, relu=True):
    x = F.conv1d(x, w[p + ".conv.weight"], None, padding=k // 2)
    x = F.relu(x) if relu else x
    return F.batch_norm(x, w[p + ".bnorm.running_mean"], w[p + ".bnorm.running_var"], w[p + ".bnorm.weight"],
                        w[p + ".bnorm.bias"], False, 0.0, 1e-5)


def _cbhg(w, p, x, K, num_highways):
    residual, T = x, x.size(-1)
    x = torch.cat([_bnconv(w, f"{p}.conv1d_bank.{k - 1}", x, k)[:, :, :T] for k in range(1, K + 1)], dim=1)
    x = F.max_pool1d(x, kernel_size=2, stride=1, padding=1)[:, :, :T]
    x = _bnconv(w, p + ".conv_project2", _bnconv(w, p + ".conv_project1", x, 3), 3, relu=False)
    x = (x + residual).transpose(1, 2)
    if p + ".pre_highway.weight" in w:
        x = F.linear(x, w[p + ".pre_highway.weight"])
    for i in range(num_highways):
        q = f"{p}.highways.{i}"
        g = torch.sigmoid(F.linear(x, w[q + ".W2.weight"], w[q + ".W2.bias"]))
        x = g * F.relu(F.linear(x, w[q + ".W1.weight"], w[q + ".W1.bias"])) + (1. - g) * x
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    flat = [w[f"{p}.rnn.{n}"] for n in names] + [w[f"{p}.rnn.{n}_reverse"] for n in names]
    h0 = torch.zeros(2, x.size(0), flat[1].shape[1], device=x.device)
    return torch.gru(x, h0, flat, True, 1, 0.0, False, True, True)[0]


def _mha(w, p, query, key, key_dim, heads, units):
    q = F.linear(query, w[p + ".W_query.weight"])
    k = F.linear(key, w[p + ".W_key.weight"])
    v = F.linear(key, w[p + ".W_value.weight"])
    sp = units // heads
    q, k, v = (torch.stack(torch.split(t, sp, dim=2), dim=0) for t in (q, k, v))
    sc = F.softmax(torch.matmul(q, k.transpose(2, 3)) / (key_dim ** 0.5), dim=3)
    out = torch.matmul(sc, v)
    return torch.cat(torch.split(out, 1, dim=0), dim=3).squeeze(0)


def _style_embed(w, hp, spk, style_idx):
    E = w["gst.stl.attention.W_query.weight"].shape[0]
    heads = E // w["gst.stl.embed"].shape[1]
    dev = spk.device
    if 0 <= style_idx < 10:
        key = torch.tanh(w["gst.stl.embed"])[style_idx].unsqueeze(0).expand(1, -1, -1)
        return _mha(w, "gst.stl.attention", torch.zeros(1, 1, E, device=dev), key, E // heads, heads, E)
    B = spk.size(0)
    out = torch.zeros(B, 1, hp.speaker_embedding_size, device=dev).view(B, 1, -1, 256)
    i = 0
    while f"gst.encoder.convs.{i}.weight" in w:
        out = F.conv2d(out, w[f"gst.encoder.convs.{i}.weight"], w[f"gst.encoder.convs.{i}.bias"], stride=2, padding=1)
        out = F.relu(F.batch_norm(out, w[f"gst.encoder.bns.{i}.running_mean"], w[f"gst.encoder.bns.{i}.running_var"],
                                  w[f"gst.encoder.bns.{i}.weight"], w[f"gst.encoder.bns.{i}.bias"], False, 0.0, 1e-5))
        i += 1
    out = out.transpose(1, 2)
    out = out.contiguous().view(B, out.size(1), -1)
    h = torch.zeros(B, E // 2, device=dev)
    for t in range(out.size(1)):
        h = torch.gru_cell(out[:, t], h, w["gst.encoder.gru.weight_ih_l0"], w["gst.encoder.gru.weight_hh_l0"],
                           w["gst.encoder.gru.bias_ih_l0"], w["gst.encoder.gru.bias_hh_l0"])
    keys = torch.tanh(w["gst.stl.embed"]).unsqueeze(0).expand(B, -1, -1)
    return _mha(w, "gst.stl.attention", torch.cat([h, spk], dim=-1).unsqueeze(1), keys, E // heads, heads, E)


@torch.no_grad()
def encoder_memory(w, hp, chars, spk, style_idx, masks=None):
    """-> (encoder_seq [B,T,P], encoder_seq_proj [B,T,D]) on chars.device."""
    x = F.embedding(chars, w["encoder.embedding.weight"])
    x = _prenet(w, "encoder.pre_net", x, hp.tts_dropout, masks).transpose(1, 2)
    x = _cbhg(w, "encoder.cbhg", x, hp.tts_encoder_K, hp.tts_num_highways)
    B, T = x.size(0), x.size(1)
    e = spk.repeat_interleave(T, dim=1).reshape(B, spk.size(1), T).transpose(1, 2)  # tacotron.py:187-193
    x = torch.cat((x, e), 2)
    if "gst.stl.embed" in w:
        x = torch.cat([x, _style_embed(w, hp, spk, style_idx).expand(B, T, -1)], dim=-1)
    return x.contiguous(), F.linear(x, w["encoder_proj.weight"]).contiguous()
