"""GST style embedding (tacotron.py:243-252, global_style_token.py): a [1|B, E] vector per chunk
from ten 64-d tokens / a 1-step GRU -- a few kFLOP, computed with tensor ops on the device and
handed to the HIP text encoder (mb_taco_encode), which does everything else of
tacotron.py:234-255 (embedding, PreNet, encoder CBHG, speaker/style concat, encoder_proj)."""
import torch
import torch.nn.functional as F


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
def style_embed(w, hp, spk, style_idx):
    """-> [1 or B, E] style embedding (or None when the checkpoint has no GST)."""
    if "gst.stl.embed" not in w:
        return None
    e = _style_embed(w, hp, spk, style_idx)
    return e.reshape(e.shape[0], -1).contiguous()
