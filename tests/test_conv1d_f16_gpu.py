"""Parity of the fp16 MFMA conv primitive (mb_conv1d_f16: fp16 storage, fp32 accumulate, time-major
activations) against ATen CPU convs evaluated on the SAME fp16-rounded operands in fp32 -- so the
only differences are summation order and the final fp16 rounding of the stored result.
Gate: max|delta| <= 2^-10 * max(1, |ref|) (one fp16 ulp of the stored value) + 1e-3 slack."""
import pytest
import torch
import torch.nn.functional as F

import hiputil

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def _h(t):
    return t.half().float()


def _check(y, ref, what, out_fp16=True):
    e = hiputil.relerr(y, ref)
    d = (y.double().cpu() - ref.double()).abs()
    tol = 1e-3 + (2.0 ** -10 if out_fp16 else 1e-4) * ref.double().abs().clamp(min=1.0)
    bad = int((d > tol).sum())
    assert e["nan"] == 0 and bad == 0, (what, e, bad)


CONV_CASES = [
    # (B, Cin, Cout, T, k, dil) -- HiFi-GAN / Fre-GAN layer shapes, all three kernel instances
    (1, 80, 512, 40, 7, 1),      # conv_pre (Cin = 5 x 16, 16 channel tiles -> 2x2 waves, grid.y = 4)
    (2, 256, 256, 200, 3, 1),    # 4 channel chunks of 64
    (1, 256, 256, 130, 11, 5),   # widest HiFi-GAN halo
    (1, 128, 128, 333, 7, 3),
    (1, 64, 64, 700, 11, 7),     # Fre-GAN dilation 7; MT=2 x 1 wave row, time tiles of 512
    (3, 32, 32, 1000, 3, 5),     # MT=1
    (1, 16, 16, 900, 11, 1),     # Fre-GAN last stage (half an MFMA tile of channels)
    (1, 128, 128, 37, 1, 1),     # 1x1
    (1, 48, 96, 50, 3, 1),       # Cin = 3 x 16 (ragged chunk), Cout = 3 tiles (inactive 4th)
    (1, 24, 40, 77, 5, 2),       # Cin % 16 = 8 (zero-filled half block), Cout % 32 = 8
    (3, 80, 512, 201, 7, 1),     # conv_pre on the short-conv kernel: two groups of 8 channel tiles, 35 k-steps padded to 40
    (2, 192, 512, 100, 7, 1),    # VITS conv_pre (84 k-steps -> 88)
    (2, 64, 256, 90, 3, 1),      # 8 channel tiles, 12 k-steps -> 16
]


@pytest.mark.parametrize("B,Cin,Cout,T,k,dil", CONV_CASES)
def test_conv1d_f16_same(cuda, lib, B, Cin, Cout, T, k, dil):
    x = _rand(B, Cin, T, seed=1)
    w = _rand(Cout, Cin, k, seed=2) / (Cin * k) ** 0.5
    b = _rand(Cout, seed=3)
    pad = (k * dil - dil) // 2
    ref = F.conv1d(_h(x), _h(w), b, dilation=dil, padding=pad)
    y = hiputil.conv1d_f16_hip(x, w, b, dilation=dil, pad=pad)
    assert y.shape == ref.shape
    _check(y, ref, "conv")


@pytest.mark.parametrize("B,Cin,Cout,T,k,u", [
    (1, 512, 256, 40, 10, 5), (2, 256, 128, 100, 10, 5), (1, 128, 64, 300, 8, 4), (1, 64, 32, 777, 4, 2),
    (1, 32, 16, 1100, 4, 2),   # Fre-GAN ups[4]
    (1, 80, 256, 33, 10, 5),   # Fre-GAN cond_up[0]
])
def test_conv_transpose1d_f16(cuda, lib, B, Cin, Cout, T, k, u):
    x = _rand(B, Cin, T, seed=6)
    w = _rand(Cin, Cout, k, seed=7) / (Cin * k / u) ** 0.5
    b = _rand(Cout, seed=8)
    pad = u // 2 + u % 2
    ref = F.conv_transpose1d(_h(x), _h(w), b, stride=u, padding=pad, output_padding=u % 2)
    y = hiputil.conv1d_f16_hip(x, w, b, transposed=True, up=u, pad=pad)
    assert y.shape == ref.shape
    _check(y, ref, "convT")


def test_f16_fused_prologue_epilogue(cuda, lib):
    # y_old + ((conv(lrelu(x)) + b) + res) * out_scale -- the resblock tail (models.py:38-43,140-145)
    x, res, yold = _rand(2, 64, 300, seed=9), _rand(2, 64, 300, seed=10), _rand(2, 64, 300, seed=11)
    w, b = _rand(64, 64, 7, seed=12) / 21.0, _rand(64, seed=13)
    ref = _h(yold) + (F.conv1d(F.leaky_relu(_h(x), float(torch.tensor(0.1).half())), _h(w), b, padding=3) + _h(res)) * (1.0 / 3.0)
    y = hiputil.conv1d_f16_hip(x, w, b, pad=3, in_act=1, in_slope=0.1, res=res, out_scale=1.0 / 3.0,
                               accumulate_into=yold)
    _check(y, ref, "fused")


def test_f16_conv_post_tanh_f32_out(cuda, lib):
    # conv_post: Cout = 1, leaky_relu(0.01) prologue, tanh, fp32 waveform out (models.py:146-148)
    for cin, T in ((32, 800), (16, 513)):
        x = _rand(2, cin, T, seed=14)
        w, b = _rand(1, cin, 7, seed=15) / (cin * 7) ** 0.5, _rand(1, seed=16)
        xa = torch.maximum(_h(x), _h(_h(x) * float(torch.tensor(0.01).half())))  # packed-half leaky relu
        ref = torch.tanh(F.conv1d(xa, _h(w), b, padding=3))
        y = hiputil.conv1d_f16_hip(x, w, b, pad=3, in_act=1, in_slope=0.01, out_act=2, y_f32=True)
        assert y.shape == ref.shape
        _check(y, ref, "conv_post", out_fp16=False)


def test_f16_nearest_repeat(cuda, lib):
    # nearest-neighbour upsample x2 folded into a 1x1 conv (fregan/generator.py:104-110)
    x = _rand(2, 128, 75, seed=23)
    w, b = _rand(64, 128, 1, seed=24) / 11.3, _rand(64, seed=25)
    ref = F.conv1d(F.interpolate(_h(x), scale_factor=2, mode="nearest"), _h(w), b)
    y = hiputil.conv1d_f16_hip(x, w, b, in_repeat=2)
    assert y.shape == ref.shape
    _check(y, ref, "repeat")


def test_layout_converters_roundtrip(cuda, lib):
    import ctypes as C
    from mockingbird_amd import _lib
    L = _lib.lib()
    x = _rand(3, 80, 57, seed=30).cuda()
    h = torch.empty(3, 57, 80, dtype=torch.float16, device="cuda")
    _lib.check(L.mb_f32_to_f16_tm(x.data_ptr(), h.data_ptr(), 3, 80, 57, _lib.stream_ptr()))
    back = torch.empty(3, 80, 57, device="cuda")
    _lib.check(L.mb_f16_tm_to_f32(h.data_ptr(), back.data_ptr(), 3, 80, 57, _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(h, x.transpose(1, 2).contiguous().half())
    assert torch.equal(back, x.half().float())


def test_short_conv_and_conv_post_kernels_equal_general_kernel(cuda, lib, monkeypatch):
    # convt_f16_kernel (up = 1) / conv_c1_f16_kernel against the general MFMA kernel (MBHIP_CONVT_GENERAL=1, read per call)
    x = _rand(2, 80, 333, seed=31)
    w, b = _rand(512, 80, 7, seed=32) / 23.7, _rand(512, seed=33)
    xp = _rand(3, 32, 2049, seed=34)
    wp, bp = _rand(1, 32, 7, seed=35) / 15.0, _rand(1, seed=36)
    y_new = hiputil.conv1d_f16_hip(x, w, b, pad=3)
    p_new = hiputil.conv1d_f16_hip(xp, wp, bp, pad=3, in_act=1, in_slope=0.01, out_act=2, y_f32=True)
    monkeypatch.setenv("MBHIP_CONVT_GENERAL", "1")
    y_old = hiputil.conv1d_f16_hip(x, w, b, pad=3)
    p_old = hiputil.conv1d_f16_hip(xp, wp, bp, pad=3, in_act=1, in_slope=0.01, out_act=2, y_f32=True)
    assert float((y_new.float() - y_old.float()).abs().max()) <= 4e-3   # one fp16 ulp at |y| < 4 (summation order)
    assert float((p_new - p_old).abs().max()) <= 2e-6


@pytest.mark.parametrize("B,Cin,Cout,T,u,with_res", [(2, 128, 64, 301, 2, True), (3, 64, 32, 1000, 2, True), (2, 32, 16, 777, 2, True),
                                                     (1, 32, 16, 130, 1, False), (2, 64, 32, 95, 3, False)])
def test_pointwise_repeat_residual_kernel(cuda, lib, monkeypatch, B, Cin, Cout, T, u, with_res):
    # conv_pw_f16_kernel: 1x1 conv over the nearest-repeated input + residual (fregan/generator.py:104-110,145-159)
    x = _rand(B, Cin, T, seed=41)
    w, b = _rand(Cout, Cin, 1, seed=42) / Cin ** 0.5, _rand(Cout, seed=43)
    res = _rand(B, Cout, T * u, seed=44) if with_res else None
    xr = F.interpolate(_h(x), scale_factor=u, mode="nearest") if u > 1 else _h(x)
    ref = F.conv1d(xr, _h(w), b)
    if with_res:
        ref = _h(ref) + _h(res)   # the conv result is rounded to fp16 before the (fp16) residual add
    y = hiputil.conv1d_f16_hip(x, w, b, in_repeat=u, res=res)
    assert y.shape == ref.shape
    _check(y, ref, "pointwise")
    monkeypatch.setenv("MBHIP_CONVT_GENERAL", "1")
    y_old = hiputil.conv1d_f16_hip(x, w, b, in_repeat=u, res=res)
    assert float((y.float() - y_old.float()).abs().max()) <= 8e-3


def test_conv_post_16_channels(cuda, lib):
    # Fre-GAN's conv_post (16 -> 1 channels) on the dot-product kernel
    x = _rand(2, 16, 3001, seed=51)
    w, b = _rand(1, 16, 7, seed=52) / (16 * 7) ** 0.5, _rand(1, seed=53)
    xa = torch.maximum(_h(x), _h(_h(x) * float(torch.tensor(0.01).half())))
    ref = torch.tanh(F.conv1d(xa, _h(w), b, padding=3))
    y = hiputil.conv1d_f16_hip(x, w, b, pad=3, in_act=1, in_slope=0.01, out_act=2, y_f32=True)
    _check(y, ref, "conv_post16", out_fp16=False)
