"""Checkpoint -> ABI weight lists.

Accepts the reference's exact state-dict layouts (SURVEY.md section 5 "Checkpoint /
resume") and produces, for each model, the ordered list of contiguous float32
CPU tensors that the matching ``mb_*_create`` entry point documents in
include/mbhip.h.  All folding (weight-norm, eval BatchNorm) is done here or in
the C library at load time, never per call.
"""
from typing import Dict, List

import torch

from . import _lib


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to("cpu", torch.float32).contiguous()


def fold_weight_norm(state: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """`weight` of a (possibly weight-normed) conv: w = g * v / ||v||, norm over
    every dim but 0 -- what torch's remove_weight_norm() leaves behind
    (models/vocoder/hifigan/models.py:152-162), for Conv1d and ConvTranspose1d alike."""
    if prefix + ".weight" in state:
        return _f32(state[prefix + ".weight"])
    for g_key, v_key in ((".weight_g", ".weight_v"),
                         (".parametrizations.weight.original0", ".parametrizations.weight.original1")):
        if prefix + g_key in state:
            g = state[prefix + g_key].detach().to("cpu", torch.float32)
            v = state[prefix + v_key].detach().to("cpu", torch.float32)
            # torch._weight_norm: v * (g / norm_except_dim(v, 2, 0))
            norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
            return (v * (g / norm)).contiguous()
    raise KeyError(f"no weight for '{prefix}' in checkpoint")


# ---------------------------------------------------------------- GAN vocoders
def gan_config(h: dict, kind: int, top_k: int = 4) -> "_lib.GanConfig":
    """AttrDict/json config (hifigan/config_16k_.json, fregan/config.json) -> mb_gan_config."""
    c = _lib.GanConfig()
    # resblock = ResBlock1 if h.resblock == '1' else ResBlock2  (hifigan/models.py:100, vits.py:251)
    c.resblock_type = 1 if str(h.get("resblock", "1")) == "1" else 2
    c.kind = kind
    # h.sampling_rate == 24000 swaps every ConvTranspose1d for Interpolate(nearest)+Conv1d (models.py:107-118)
    c.interp_ups = int(kind == 0 and int(h.get("sampling_rate", 16000)) == 24000)
    c.num_mels = int(h.get("num_mels", 80))
    c.upsample_initial_channel = int(h["upsample_initial_channel"])
    ur, uk = list(h["upsample_rates"]), list(h["upsample_kernel_sizes"])
    c.num_upsamples = len(ur)
    for i, (u, k) in enumerate(zip(ur, uk)):
        c.upsample_rates[i] = int(u)
        c.upsample_kernel_sizes[i] = int(k)
    rk, rd = list(h["resblock_kernel_sizes"]), list(h["resblock_dilation_sizes"])
    c.num_kernels = len(rk)
    c.num_dilations = len(rd[0])
    for j, k in enumerate(rk):
        c.resblock_kernel_sizes[j] = int(k)
        for d, dil in enumerate(rd[j]):
            c.resblock_dilations[j][d] = int(dil)
    c.top_k = top_k
    return c


def gan_weight_list(state: Dict[str, torch.Tensor], cfg: "_lib.GanConfig") -> List[torch.Tensor]:
    up = "ups.{}.1" if cfg.interp_ups else "ups.{}"  # Sequential(InterpolationBlock, Conv1d)  models.py:108-112
    names = ["conv_pre"] + [up.format(i) for i in range(cfg.num_upsamples)]
    if cfg.kind == 1:
        lvl = cfg.num_upsamples - cfg.top_k
        names += [f"cond_up.{i}" for i in range(cfg.num_upsamples - lvl)]
        names += [f"res_output.{i}.1" for i in range(cfg.num_upsamples - lvl - 1)]
    for i in range(cfg.num_upsamples * cfg.num_kernels):
        if cfg.resblock_type == 2:  # ResBlock2.convs (models.py:54-61): two convs, the block's first two dilations
            names += [f"resblocks.{i}.convs.{d}" for d in range(2)]
            continue
        names += [f"resblocks.{i}.convs1.{d}" for d in range(cfg.num_dilations)]
        names += [f"resblocks.{i}.convs2.{d}" for d in range(cfg.num_dilations)]
    names.append("conv_post")
    out = []
    for n in names:
        out.append(fold_weight_norm(state, n))
        out.append(_f32(state[n + ".bias"]))
    return out


def fold_batchnorm(state, prefix: str, eps: float = 1e-5):
    """Eval-mode BatchNorm1d as y = x*scale + shift."""
    w = state[prefix + ".weight"].detach().double().cpu()
    b = state[prefix + ".bias"].detach().double().cpu()
    m = state[prefix + ".running_mean"].detach().double().cpu()
    v = state[prefix + ".running_var"].detach().double().cpu()
    scale = w / torch.sqrt(v + eps)
    shift = b - m * scale
    return scale.float().contiguous(), shift.float().contiguous()


# --------------------------------------------------------------------- WaveRNN
def wavernn_config(hp) -> "_lib.WaveRNNConfig":
    """hp: module/dict with the reference's wavernn/hparams.py names."""
    g = (lambda k: hp[k]) if isinstance(hp, dict) else (lambda k: getattr(hp, k))
    c = _lib.WaveRNNConfig()
    c.rnn_dims, c.fc_dims, c.bits, c.pad = int(g("voc_rnn_dims")), int(g("voc_fc_dims")), int(g("bits")), int(g("voc_pad"))
    f = tuple(g("voc_upsample_factors"))
    c.n_upsample = len(f)
    for i, s in enumerate(f):
        c.upsample_factors[i] = int(s)
    c.feat_dims, c.compute_dims = int(g("num_mels")), int(g("voc_compute_dims"))
    c.res_out_dims, c.res_blocks = int(g("voc_res_out_dims")), int(g("voc_res_blocks"))
    mode = g("voc_mode")
    if mode not in ("RAW", "MOL"):
        raise RuntimeError(f"Unknown model mode value - {mode}")  # fatchord_version.py:98,230
    c.mode = 0 if mode == "RAW" else 1
    return c


def wavernn_weight_list(state: Dict[str, torch.Tensor], cfg: "_lib.WaveRNNConfig") -> List[torch.Tensor]:
    """ABI order of mb_wavernn_create (include/mbhip.h section 3)."""
    def bn(p):
        return [p + ".weight", p + ".bias", p + ".running_mean", p + ".running_var"]
    r = "upsample.resnet."
    names = [r + "conv_in.weight"] + bn(r + "batch_norm")
    for i in range(cfg.res_blocks):
        q = f"{r}layers.{i}."
        names += [q + "conv1.weight"] + bn(q + "batch_norm1") + [q + "conv2.weight"] + bn(q + "batch_norm2")
    names += [r + "conv_out.weight", r + "conv_out.bias"]
    names += [f"upsample.up_layers.{2 * i + 1}.weight" for i in range(cfg.n_upsample)]
    names += ["I.weight", "I.bias"]
    for n in ("rnn1", "rnn2"):
        names += [f"{n}.weight_ih_l0", f"{n}.weight_hh_l0", f"{n}.bias_ih_l0", f"{n}.bias_hh_l0"]
    names += ["fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc3.weight", "fc3.bias"]
    return [_f32(state[n]) for n in names]


# -------------------------------------------------------------------- Tacotron
def taco_config(state: Dict[str, torch.Tensor], r: int = None, max_r: int = 20, dropout: float = 0.5) -> "_lib.TacoConfig":
    """Shapes are read from the checkpoint (config must come from the loaded
    objects, not be hard-coded)."""
    c = _lib.TacoConfig()
    c.dropout = float(dropout) if dropout and dropout > 0 else -1.0  # hparams.tts_dropout: PreNet dropout stays on at inference (pre_net.py:23,26); ABI: 0 = default 0.5, < 0 = off
    fc1 = state["decoder.prenet.fc1.weight"]
    c.n_mels = fc1.shape[1]
    c.decoder_dims = fc1.shape[0] // 2
    c.lstm_dims = state["decoder.res_rnn1.weight_hh"].shape[1]
    c.project_dims = state["decoder.rnn_input.weight"].shape[1] - c.decoder_dims
    c.max_r = state["decoder.mel_proj.weight"].shape[0] // c.n_mels if max_r is None else max_r
    c.r = int(state["decoder.r"].item()) if r is None else int(r)
    c.postnet_dims = state["postnet.conv_project1.conv.weight"].shape[0]
    c.postnet_K = sum(1 for k in state if k.startswith("postnet.conv1d_bank.") and k.endswith(".conv.weight"))
    c.num_highways = sum(1 for k in state if k.startswith("postnet.highways.") and k.endswith(".W1.weight"))
    cw = state["decoder.attn_net.conv.weight"]
    c.lsa_filters, c.lsa_kernel = cw.shape[0], cw.shape[2]
    if "encoder.embedding.weight" in state:
        emb = state["encoder.embedding.weight"]
        c.has_encoder = 1
        c.num_chars, c.embed_dims = emb.shape
        c.encoder_dims = state["encoder.pre_net.fc1.weight"].shape[0]
        c.encoder_K = sum(1 for k in state if k.startswith("encoder.cbhg.conv1d_bank.") and k.endswith(".conv.weight"))
        c.style_dims = state["gst.stl.attention.W_value.weight"].shape[0] if "gst.stl.attention.W_value.weight" in state else 0
        c.speaker_dims = c.project_dims - c.encoder_dims - c.style_dims
        if c.style_dims:
            c.has_gst = 1
            emb = state["gst.stl.embed"]
            c.gst_tokens = emb.shape[0]
            c.gst_heads = c.style_dims // emb.shape[1]
            n = 0
            while f"gst.encoder.convs.{n}.weight" in state:
                c.gst_filters[n] = state[f"gst.encoder.convs.{n}.weight"].shape[0]
                n += 1
            c.gst_n_convs = n
            # ReferenceEncoder views its (all-zero, [N,1,speaker_embedding_size]) input as [N,1,-1,n_mels]
            # with GSTHyperparameters.n_mels = 256 (global_style_token.py:55, gst_hyperparameters.py:13)
            c.gst_width = 256 if c.speaker_dims % 256 == 0 else c.speaker_dims
    return c


def taco_weight_list(state: Dict[str, torch.Tensor], cfg: "_lib.TacoConfig") -> List[torch.Tensor]:
    """ABI order of mb_taco_create (decoder + postnet + post_proj)."""
    def bn(p):
        return [p + ".weight", p + ".bias", p + ".running_mean", p + ".running_var"]
    d = "decoder."
    names = [d + "prenet.fc1.weight", d + "prenet.fc1.bias", d + "prenet.fc2.weight", d + "prenet.fc2.bias",
             d + "attn_net.conv.weight", d + "attn_net.conv.bias", d + "attn_net.L.weight", d + "attn_net.W.weight",
             d + "attn_net.W.bias", d + "attn_net.v.weight",
             d + "attn_rnn.weight_ih", d + "attn_rnn.weight_hh", d + "attn_rnn.bias_ih", d + "attn_rnn.bias_hh",
             d + "rnn_input.weight", d + "rnn_input.bias"]
    for n in ("res_rnn1", "res_rnn2"):
        names += [f"{d}{n}.weight_ih", f"{d}{n}.weight_hh", f"{d}{n}.bias_ih", f"{d}{n}.bias_hh"]
    names += [d + "mel_proj.weight", d + "stop_proj.weight", d + "stop_proj.bias"]
    p = "postnet."
    for k in range(cfg.postnet_K):
        names += [f"{p}conv1d_bank.{k}.conv.weight"] + bn(f"{p}conv1d_bank.{k}.bnorm")
    names += [p + "conv_project1.conv.weight"] + bn(p + "conv_project1.bnorm")
    names += [p + "conv_project2.conv.weight"] + bn(p + "conv_project2.bnorm")
    names += [p + "pre_highway.weight"]
    for i in range(cfg.num_highways):
        names += [f"{p}highways.{i}.W1.weight", f"{p}highways.{i}.W1.bias", f"{p}highways.{i}.W2.weight", f"{p}highways.{i}.W2.bias"]
    for sfx in ("", "_reverse"):
        names += [f"{p}rnn.weight_ih_l0{sfx}", f"{p}rnn.weight_hh_l0{sfx}", f"{p}rnn.bias_ih_l0{sfx}", f"{p}rnn.bias_hh_l0{sfx}"]
    names += ["post_proj.weight"]
    if cfg.has_encoder:
        e = "encoder."
        names += [e + "embedding.weight", e + "pre_net.fc1.weight", e + "pre_net.fc1.bias", e + "pre_net.fc2.weight",
                  e + "pre_net.fc2.bias"]
        q = e + "cbhg."
        for k in range(cfg.encoder_K):
            names += [f"{q}conv1d_bank.{k}.conv.weight"] + bn(f"{q}conv1d_bank.{k}.bnorm")
        names += [q + "conv_project1.conv.weight"] + bn(q + "conv_project1.bnorm")
        names += [q + "conv_project2.conv.weight"] + bn(q + "conv_project2.bnorm")
        if q + "pre_highway.weight" in state:
            names += [q + "pre_highway.weight"]
        for i in range(cfg.num_highways):
            names += [f"{q}highways.{i}.W1.weight", f"{q}highways.{i}.W1.bias", f"{q}highways.{i}.W2.weight", f"{q}highways.{i}.W2.bias"]
        for sfx in ("", "_reverse"):
            names += [f"{q}rnn.weight_ih_l0{sfx}", f"{q}rnn.weight_hh_l0{sfx}", f"{q}rnn.bias_ih_l0{sfx}", f"{q}rnn.bias_hh_l0{sfx}"]
        names += ["encoder_proj.weight"]
        if cfg.has_gst:
            g = "gst.encoder."
            for i in range(cfg.gst_n_convs):
                names += [f"{g}convs.{i}.weight", f"{g}convs.{i}.bias"] + bn(f"{g}bns.{i}")
            names += [g + "gru.weight_ih_l0", g + "gru.weight_hh_l0", g + "gru.bias_ih_l0", g + "gru.bias_hh_l0",
                      "gst.stl.embed", "gst.stl.attention.W_query.weight", "gst.stl.attention.W_key.weight",
                      "gst.stl.attention.W_value.weight"]
    return [_f32(state[n]) for n in names]
