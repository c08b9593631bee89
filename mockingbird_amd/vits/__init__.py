"""VITS decoder: Generator.forward of
models/synthesizer/models/vits.py:245-298 -- the HiFi-GAN-style stack that turns the latent z (and the
speaker embedding g) into the waveform inside SynthesizerTrn.infer (`o = self.dec(z * y_mask, g=g)`).

It runs on the same kernels as the HiFi-GAN vocoder (mb_gan_* with num_mels = the latent width; fp32 MFMA
or the fp16 fused-ResBlock path); what differs is handled here / by mb_gan_forward_ex:
  * conv_pre is a plain Conv1d (no weight norm), conv_post has no bias (zeros are passed),
  * `x = conv_pre(x) + cond(g)`: cond is a 1x1 conv of the [B, gin, 1] speaker embedding, i.e. a
    per-utterance channel bias; it is evaluated with mb_conv1d and handed to mb_gan_forward_ex,
  * ConvTranspose1d padding (k-u)//2 equals HiFi-GAN's u//2 + u%2 for the shipped k = 2u, u even configs
    (checked at construction).
There is no CPU path."""
import ctypes as C

import torch

from .. import _lib
from ..vocoder.gan import GanGenerator, KIND_HIFIGAN


class VitsGenerator:
    def __init__(self, state_dict, initial_channel, resblock, resblock_kernel_sizes, resblock_dilation_sizes,
                 upsample_rates, upsample_initial_channel, upsample_kernel_sizes, gin_channels=0, dtype="f32"):
        """Same positional signature as the reference Generator.__init__ (vits.py:246), preceded by the
        `dec.*` state_dict of a SynthesizerTrn checkpoint (keys without the `dec.` prefix)."""
        for u, k in zip(upsample_rates, upsample_kernel_sizes):
            if (k - u) // 2 != u // 2 + u % 2 or u % 2:
                raise _lib.MbHipError(f"VITS upsample (rate {u}, kernel {k}): only k = 2u with even u is implemented")
        h = dict(num_mels=initial_channel, resblock=str(resblock), resblock_kernel_sizes=list(resblock_kernel_sizes),
                 resblock_dilation_sizes=[list(d) for d in resblock_dilation_sizes], upsample_rates=list(upsample_rates),
                 upsample_initial_channel=upsample_initial_channel, upsample_kernel_sizes=list(upsample_kernel_sizes),
                 sampling_rate=0)
        st = dict(state_dict)
        st["conv_post.bias"] = torch.zeros(1)  # Conv1d(ch, 1, 7, bias=False)  vits.py:266
        self.gen = GanGenerator(h, st, KIND_HIFIGAN, dtype=dtype)
        self.hop = self.gen.hop
        self.gin_channels = gin_channels
        self.c0 = upsample_initial_channel
        self._cond = None
        if gin_channels:
            L = _lib.lib()
            w = st["cond.weight"].detach().to(torch.float32).contiguous().cpu()  # [C0, gin, 1]
            n = L.mb_conv1d_packed_floats(self.c0, gin_channels, 1, 1)
            packed = torch.empty(n, dtype=torch.float32)
            _lib.check(L.mb_conv1d_pack(w.data_ptr(), self.c0, gin_channels, 1, 1, 0, 0, packed.data_ptr()), "mb_conv1d_pack")
            self._cond = (packed.cuda(), st["cond.bias"].detach().to(torch.float32).contiguous().cuda())

    def _cond_bias(self, g):
        """cond(g): [B, gin, 1] -> [B, C0] through the fp32 MFMA conv primitive (a 1x1 conv over one position)."""
        B = g.shape[0]
        g = g.to(torch.float32).reshape(B, self.gin_channels, 1).contiguous()
        y = torch.empty(B, self.c0, 1, device=g.device)
        a = _lib.ConvArgs()
        a.d_x, a.d_wpacked, a.d_bias, a.d_y = g.data_ptr(), self._cond[0].data_ptr(), self._cond[1].data_ptr(), y.data_ptr()
        a.x_bstride, a.y_bstride, a.res_bstride = self.gin_channels, self.c0, self.c0
        a.batch, a.c_in, a.c_out, a.t_in, a.t_out = B, self.gin_channels, self.c0, 1, 1
        a.ksize, a.dilation, a.pad, a.up = 1, 1, 0, 1
        a.in_scale, a.out_scale, a.in_repeat = 1.0, 1.0, 1
        _lib.check(_lib.lib().mb_conv1d(C.byref(a), _lib.stream_ptr()), "mb_conv1d")
        return y.reshape(B, self.c0)

    def forward(self, x, g=None):
        """x [B, initial_channel, T] (CUDA), g [B, gin, 1] or None -> [B, 1, T*hop]."""
        if not x.is_cuda:
            raise _lib.MbHipError("VitsGenerator.forward needs a CUDA(HIP) tensor; there is no CPU path")
        bias = None
        if g is not None:
            if self._cond is None:
                raise _lib.MbHipError("this decoder was built with gin_channels=0")
            bias = self._cond_bias(g.to(x.device))
        return self.gen.forward(x, chan_bias=bias)

    __call__ = forward
