"""ppg2mel voice-conversion decoder: the autoregressive loop of
models/ppg2mel/rnn_decoder_mol.py:Decoder.inference / inference_batched on HIP kernels
(csrc/ppg2mel.hip, mb_ppg2mel_*).  `Ppg2MelDecoder` mirrors the reference Decoder's inference-time
surface: `inference(memory, stop_threshold=0.5)` and `inference_batched(memory, stop_threshold=0.5)` with
the reference's return values; it is built from the `decoder.*` entries of a MelDecoderMOLv2 checkpoint.
`MelDecoderMOLv2` mirrors the whole reference model at inference time (models/ppg2mel/__init__.py:20-192,
`inference(bottle_neck_features, logf0_uv, spembs)`): the convolutional front end (bnf_prenet, pitch_convs),
the speaker projection and the CNN postnet run on csrc/ppg_net.hip (mb_ppg2mel_net_*), the decoder loop in
between on `Ppg2MelDecoder`; `load_model(model_file, device)` is the reference's loader of the same name.
There is no CPU path."""
import ctypes as C

import numpy as np
import torch

from .. import _lib

DEFAULT_HP = dict(enc_dim=256, num_mels=80, frames_per_step=2, attention_rnn_dim=512, decoder_rnn_dim=512,
                  prenet_dims=(256, 128), num_mixtures=5, encoder_down_factor=4, num_decoder_rnn_layer=1,
                  concat_context_to_last=True)


def weight_list(state, hp):
    """state_dict of the reference Decoder module -> tensors in the ABI order (include/mbhip.h section 6)."""
    names = [f"prenet.layers.{i}.linear_layer.weight" for i in range(len(hp["prenet_dims"]))]
    names += ["attention_rnn.weight_ih", "attention_rnn.weight_hh", "attention_rnn.bias_ih", "attention_rnn.bias_hh",
              "attention_layer.query_layer.0.weight", "attention_layer.query_layer.0.bias",
              "attention_layer.query_layer.2.weight", "attention_layer.query_layer.2.bias"]
    for i in range(hp["num_decoder_rnn_layer"]):
        names += [f"decoder_rnn_layers.{i}.{n}" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    names += ["linear_projection.linear_layer.weight", "linear_projection.linear_layer.bias",
              "stop_layer.linear_layer.weight", "stop_layer.linear_layer.bias"]
    return [state[n].detach().to(torch.float32).contiguous().cpu() for n in names]


class Ppg2MelDecoder:
    def __init__(self, state_dict, hp=None):
        if not torch.cuda.is_available():
            raise _lib.MbHipError("ppg2mel decoder: no MI355X visible; this build has no CPU path")
        self.hp = dict(DEFAULT_HP if hp is None else hp)
        h = self.hp
        cfg = _lib.Ppg2MelConfig()
        cfg.enc_dim, cfg.num_mels, cfg.frames_per_step = h["enc_dim"], h["num_mels"], h["frames_per_step"]
        cfg.attention_rnn_dim, cfg.decoder_rnn_dim = h["attention_rnn_dim"], h["decoder_rnn_dim"]
        cfg.n_prenet = len(h["prenet_dims"])
        for i, d in enumerate(h["prenet_dims"]):
            cfg.prenet_dims[i] = d
        cfg.num_mixtures, cfg.encoder_down_factor = h["num_mixtures"], h["encoder_down_factor"]
        cfg.num_decoder_rnn_layer, cfg.concat_context_to_last = h["num_decoder_rnn_layer"], int(h["concat_context_to_last"])
        self.cfg = cfg
        L = _lib.lib()
        ws = weight_list(state_dict, h)
        n = L.mb_ppg2mel_num_weights(C.byref(cfg))
        if n != len(ws):
            raise _lib.MbHipError(f"ppg2mel: ABI expects {n} weight tensors, checkpoint mapping gives {len(ws)}")
        for i, w in enumerate(ws):
            want = L.mb_ppg2mel_weight_numel(C.byref(cfg), i)
            if w.numel() != want:
                raise _lib.MbHipError(f"ppg2mel weight {i}: {tuple(w.shape)} has {w.numel()} elements, expected {want}")
        arr = _lib.host_ptr_array(ws)
        hnd = C.c_void_p()
        _lib.check(L.mb_ppg2mel_create(C.byref(cfg), arr, len(ws), C.byref(hnd)), "mb_ppg2mel_create")
        self._h = hnd
        self._ws = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().mb_ppg2mel_destroy(h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    def decode(self, memory, stop_threshold=0.5, dropout=None, seed=None, max_steps=None):
        """Raw loop: memory [B, T_enc, enc_dim] (CUDA) -> (mel [B, steps, r*num_mels], alignments [B, steps, T_enc],
        stop logits [B, steps]), untruncated.  dropout: optional list of keep masks in program order
        (step-major, prenet layer inside) as produced for the oracle."""
        if not memory.is_cuda:
            raise _lib.MbHipError("ppg2mel decoder needs a CUDA(HIP) tensor; there is no CPU path")
        if seed is None:  # the reference's prenet dropout draws from torch's global generator
            seed = _lib.fresh_seed()
        memory = memory.to(torch.float32).contiguous()
        B, T, E = memory.shape
        h = self.hp
        if E != h["enc_dim"]:
            raise _lib.MbHipError(f"memory has enc_dim {E}, model expects {h['enc_dim']}")
        r, nm = h["frames_per_step"], h["num_mels"]
        lim = T * h["encoder_down_factor"] // r  # rnn_decoder_mol.py:281-282
        max_step = lim if max_steps is None else max_steps
        min_step = lim - 5
        dev = memory.device
        L = _lib.lib()
        need = L.mb_ppg2mel_workspace_bytes(self._h, B)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        mel = torch.empty(B, max_step, r * nm, device=dev)
        align = torch.empty(B, max_step, T, device=dev)
        stop = torch.empty(B, max_step, device=dev)
        dmask = None
        if dropout is not None:
            dims = list(h["prenet_dims"])
            n_steps = len(dropout) // len(dims)
            if n_steps < max_step:
                raise _lib.MbHipError(f"dropout masks cover {n_steps} steps, need {max_step}")
            for s_ in range(max_step):
                for l, d_ in enumerate(dims):
                    if tuple(dropout[s_ * len(dims) + l].shape) != (B, d_):
                        raise _lib.MbHipError(f"dropout mask {s_ * len(dims) + l} must be {(B, d_)}, got "
                                              f"{tuple(dropout[s_ * len(dims) + l].shape)}")
            per_layer = [torch.stack([dropout[s * len(dims) + l] for s in range(max_step)]) for l in range(len(dims))]
            dmask = torch.cat([p.reshape(-1) for p in per_layer]).to(dev, torch.float32).contiguous()
        n = C.c_int(0)
        _lib.check(L.mb_ppg2mel_decode(self._h, _lib.ptr(memory), B, T, max_step, min_step, float(stop_threshold),
                                       _lib.ptr(dmask), int(seed), _lib.ptr(mel), _lib.ptr(align), _lib.ptr(stop),
                                       C.byref(n), _lib.ptr(self._ws), self._ws.numel(), _lib.stream_ptr()),
                   "mb_ppg2mel_decode")
        s = n.value
        ms, st = C.c_float(), C.c_int()
        if L.mb_ppg2mel_last_loop_ms(self._h, C.byref(ms), C.byref(st)) == 0:  # production-dims step only
            self.last_loop_ms, self.last_loop_steps = ms.value, st.value
            nl = C.c_int()
            if L.mb_ppg2mel_last_loop_launches(self._h, C.byref(nl)) == 0:
                self.last_loop_launches = nl.value  # 1 = the resident loop (one utterance, csrc/ppg_resident.h)
        return mel[:, :s], align[:, :s], stop[:, :s]

    def inference(self, memory, stop_threshold=0.5, dropout=None, seed=None):
        """Decoder.inference (rnn_decoder_mol.py:267-316): memory [1, T_enc, enc_dim] ->
        (mel_outputs [1, steps*r, num_mels], alignments [1, steps, T_enc])."""
        if memory.shape[0] != 1:
            raise _lib.MbHipError("Decoder.inference takes one utterance; use inference_batched")
        mel, al, _ = self.decode(memory, stop_threshold, dropout, seed)
        return mel.reshape(1, -1, self.hp["num_mels"]), al

    def inference_batched(self, memory, stop_threshold=0.5, dropout=None, seed=None):
        """Decoder.inference_batched (rnn_decoder_mol.py:318-374): every utterance is cut at its first step whose
        sigmoid(stop) exceeds the threshold and the pieces are concatenated -> (mel [1, sum, num_mels],
        alignments [B, steps, T_enc]).  Like the reference this raises IndexError for an utterance that never
        crosses the threshold."""
        mel, al, stop = self.decode(memory, stop_threshold, dropout, seed)
        B = mel.shape[0]
        melf = mel.reshape(B, -1, self.hp["num_mels"])
        parts = []
        sg = torch.sigmoid(stop).cpu()
        for b in range(B):
            idx = np.argwhere(sg[b] > stop_threshold)[0][0].item()
            parts.append(melf[b, :idx, :])
        return torch.cat(parts, dim=0).unsqueeze(0), al


NET_DEFAULT = dict(encoder_dim=256, encoder_downsample_rates=(2, 2), attention_rnn_dim=512, decoder_rnn_dim=512,
                   num_decoder_rnn_layer=1, concat_context_to_last=True, prenet_dims=(256, 128), num_mixtures=5,
                   frames_per_step=2)


def net_weight_list(state, n_post):
    """MelDecoderMOLv2 state_dict -> the host tensors of mb_ppg2mel_net_create (include/mbhip.h section 6b)."""
    names = []
    for br in ("bnf_prenet", "pitch_convs"):
        names += [f"{br}.0.weight", f"{br}.3.weight", f"{br}.3.bias", f"{br}.6.weight", f"{br}.6.bias"]
    names += ["reduce_proj.weight", "reduce_proj.bias"]
    for i in range(n_post):
        p = f"postnet.convolutions.{i}"
        names += [p + ".0.conv.weight", p + ".0.conv.bias", p + ".1.weight", p + ".1.bias", p + ".1.running_mean",
                  p + ".1.running_var"]
    return [state[n].detach().to(torch.float32).contiguous().cpu() for n in names]


class MelDecoderMOLv2:
    """Inference-time mirror of models/ppg2mel/__init__.py:MelDecoderMOLv2 with the reference's constructor arguments
    (the `model:` section of the checkpoint's yaml).  Either pass `state_dict=` or call `load_state_dict()` afterwards,
    as the reference's callers do (run.py: `MelDecoderMOLv2(**cfg["model"]).to(device)`, `.load_state_dict(ckpt["model"])`,
    `.eval()`); the device handles are built when the weights arrive."""

    def __init__(self, num_speakers=None, spk_embed_dim=256, bottle_neck_feature_dim=144, encoder_dim=256,
                 encoder_downsample_rates=(2, 2), attention_rnn_dim=512, decoder_rnn_dim=512, num_decoder_rnn_layer=1,
                 concat_context_to_last=True, prenet_dims=(256, 128), num_mixtures=5, frames_per_step=2,
                 mask_padding=True, state_dict=None):
        if not torch.cuda.is_available():
            raise _lib.MbHipError("ppg2mel: no MI355X visible; this build has no CPU path")
        rates = [int(r) for r in encoder_downsample_rates]
        if len(rates) != 2:
            raise _lib.MbHipError("ppg2mel: the reference model has exactly two downsampling convolutions")
        self.num_mels, self.frames_per_step = 80, frames_per_step
        self.encoder_down_factor = int(np.cumprod(rates)[-1])
        self._args = dict(rates=rates, spk_embed_dim=spk_embed_dim, bottle_neck_feature_dim=bottle_neck_feature_dim,
                          encoder_dim=encoder_dim, attention_rnn_dim=attention_rnn_dim, decoder_rnn_dim=decoder_rnn_dim,
                          num_decoder_rnn_layer=num_decoder_rnn_layer, concat_context_to_last=concat_context_to_last,
                          prenet_dims=tuple(prenet_dims), num_mixtures=num_mixtures)
        self._h, self._ws, self.decoder, self.cfg = None, None, None, None
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, state_dict, strict=True):
        a = self._args
        n_post = 0
        while f"postnet.convolutions.{n_post}.0.conv.weight" in state_dict:
            n_post += 1
        pw = state_dict["postnet.convolutions.0.0.conv.weight"]
        cfg = _lib.Ppg2MelNetConfig()
        cfg.bnf_dim, cfg.spk_dim, cfg.enc_dim = a["bottle_neck_feature_dim"], a["spk_embed_dim"], a["encoder_dim"]
        cfg.down0, cfg.down1, cfg.num_mels = a["rates"][0], a["rates"][1], self.num_mels
        cfg.postnet_layers, cfg.postnet_dim, cfg.postnet_ksize = n_post, pw.shape[0], pw.shape[2]
        L = _lib.lib()
        ws = net_weight_list(state_dict, n_post)
        n = L.mb_ppg2mel_net_num_weights(C.byref(cfg))
        if n != len(ws):
            raise _lib.MbHipError(f"ppg2mel net: ABI expects {n} weight tensors, checkpoint mapping gives {len(ws)}")
        for i, w in enumerate(ws):
            want = L.mb_ppg2mel_net_weight_numel(C.byref(cfg), i)
            if w.numel() != want:
                raise _lib.MbHipError(f"ppg2mel net weight {i}: {tuple(w.shape)} has {w.numel()} elements, expected {want}")
        hnd = C.c_void_p()
        _lib.check(L.mb_ppg2mel_net_create(C.byref(cfg), _lib.host_ptr_array(ws), len(ws), C.byref(hnd)),
                   "mb_ppg2mel_net_create")
        self.__del__()  # a second load replaces the first
        self.cfg, self._h, self._ws = cfg, hnd, None
        dhp = dict(enc_dim=a["encoder_dim"], num_mels=self.num_mels, frames_per_step=self.frames_per_step,
                   attention_rnn_dim=a["attention_rnn_dim"], decoder_rnn_dim=a["decoder_rnn_dim"], prenet_dims=a["prenet_dims"],
                   num_mixtures=a["num_mixtures"], encoder_down_factor=self.encoder_down_factor,
                   num_decoder_rnn_layer=a["num_decoder_rnn_layer"], concat_context_to_last=a["concat_context_to_last"])
        self.decoder = Ppg2MelDecoder({k[len("decoder."):]: v for k, v in state_dict.items() if k.startswith("decoder.")}, dhp)
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().mb_ppg2mel_net_destroy(h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    def eval(self):
        return self

    def to(self, device):
        return self

    def _workspace(self, B, T, dev):
        need = _lib.lib().mb_ppg2mel_net_workspace_bytes(self._h, B, T)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return self._ws

    def encode(self, bottle_neck_features, logf0_uv, spembs):
        """:172-179 -> decoder memory [B, T_enc, encoder_dim]."""
        if self._h is None:
            raise _lib.MbHipError("ppg2mel: no weights yet (load_state_dict)")
        for name, t in (("bottle_neck_features", bottle_neck_features), ("logf0_uv", logf0_uv), ("spembs", spembs)):
            if t is None:
                raise AssertionError(f"{name} is required")  # the reference asserts spembs is not None
            if not t.is_cuda:
                raise _lib.MbHipError(f"ppg2mel: {name} must be a CUDA(HIP) tensor; there is no CPU path")
        bnf = bottle_neck_features.to(torch.float32).contiguous()
        lf0 = logf0_uv.to(torch.float32).contiguous()
        spk = spembs.to(torch.float32).contiguous()
        B, T, D = bnf.shape
        if D != self.cfg.bnf_dim or tuple(lf0.shape) != (B, T, 2) or tuple(spk.shape) != (B, self.cfg.spk_dim):
            raise _lib.MbHipError(f"ppg2mel: got features {tuple(bnf.shape)}, logf0_uv {tuple(lf0.shape)}, spembs "
                                  f"{tuple(spk.shape)}; expected [B, T, {self.cfg.bnf_dim}], [B, T, 2], [B, {self.cfg.spk_dim}]")
        L = _lib.lib()
        t_enc = L.mb_ppg2mel_net_t_enc(C.byref(self.cfg), T)
        if t_enc < 1:
            raise _lib.MbHipError(f"ppg2mel: {T} frames are too few for the downsampling convolutions")
        ws = self._workspace(B, T, bnf.device)
        mem = torch.empty(B, t_enc, self.cfg.enc_dim, device=bnf.device)
        _lib.check(L.mb_ppg2mel_net_encode(self._h, _lib.ptr(bnf), _lib.ptr(lf0), _lib.ptr(spk), B, T, _lib.ptr(mem),
                                           _lib.ptr(ws), ws.numel(), _lib.stream_ptr()), "mb_ppg2mel_net_encode")
        return mem

    def postnet(self, mel_outputs):
        """:186-187: mel_outputs [B, T, num_mels] -> mel_outputs + Postnet(mel_outputs)."""
        mel = mel_outputs.to(torch.float32).contiguous()
        B, T, nm = mel.shape
        if nm != self.num_mels:
            raise _lib.MbHipError(f"ppg2mel postnet: {nm} mel channels, expected {self.num_mels}")
        out = torch.empty_like(mel)
        if T == 0:
            return out
        ws = self._workspace(B, T, mel.device)
        _lib.check(_lib.lib().mb_ppg2mel_net_postnet(self._h, _lib.ptr(mel), B, T, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                                      _lib.stream_ptr()), "mb_ppg2mel_net_postnet")
        return out

    def inference(self, bottle_neck_features, logf0_uv=None, spembs=None, dropout=None, seed=None):
        """MelDecoderMOLv2.inference :166-192 -> (mel_outputs[0], mel_outputs_postnet[0], alignments[0])."""
        memory = self.encode(bottle_neck_features, logf0_uv, spembs)
        if memory.size(0) > 1:
            mel, al = self.decoder.inference_batched(memory, dropout=dropout, seed=seed)
        else:
            mel, al = self.decoder.inference(memory, dropout=dropout, seed=seed)
        return mel[0], self.postnet(mel)[0], al[0]


def load_model(model_file, device=None):
    """models/ppg2mel/__init__.py:194-209: the yaml next to the checkpoint gives the constructor arguments
    (`model:` section), the checkpoint's "model" entry the weights."""
    import yaml
    from pathlib import Path
    model_file = Path(model_file)
    cfgs = list(model_file.parent.rglob("*.yaml"))
    if len(cfgs) == 0:
        raise FileNotFoundError("No model yaml config found for convertor")
    with open(cfgs[0]) as f:
        model_cfg = yaml.safe_load(f)["model"]
    ckpt = torch.load(model_file, map_location="cpu")
    return MelDecoderMOLv2(**model_cfg).load_state_dict(ckpt["model"]).eval()
