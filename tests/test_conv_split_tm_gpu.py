"""Parity of the time-major split conv (mb_conv_split_tm, conv_split_tm.hip: conv_pre, the ConvTranspose1d upsamplers as three-tap
polyphase convs, Fre-GAN's cond_up / res_output, conv_post of the fp32 generators) against float64 ATen.
Reference: models/vocoder/hifigan/models.py:103-127,134-150; models/vocoder/fregan/generator.py:95-118,137-166.
Gate: max |delta| <= 1e-5 * max(1, output RMS) (fp32-grade sums from error-compensated fp16 products)."""
import pytest
import torch
import torch.nn.functional as F

import hiputil

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def _check(y, ref, tol=1e-5):
    rms = float(ref.pow(2).mean().sqrt())
    err = float((y.double() - ref).abs().max())
    assert int(torch.isnan(y).sum()) == 0 and err <= tol * max(1.0, rms), (err, rms)


CONVS = [
    # (B, C_in, M, T, k, in_slope): conv_pre (80 -> 512, k 7), small / ragged shapes, conv_post-like (32 -> 1), 16 input channels
    (2, 80, 512, 200, 7, 1.0), (1, 80, 256, 37, 7, 1.0), (3, 32, 1, 4000, 7, 0.01), (2, 16, 1, 3000, 7, 0.01),
    (1, 128, 64, 700, 3, 0.1), (2, 256, 640, 300, 3, 0.1), (1, 64, 96, 500, 1, 1.0), (1, 512, 1280, 130, 3, 0.1),
]


@pytest.mark.parametrize("B,Cin,M,T,k,slope", CONVS)
def test_conv_matches_float64(cuda, lib, B, Cin, M, T, k, slope):
    x = _rand(B, Cin, T, seed=1)
    w = _rand(M, Cin, k, seed=2) / (Cin * k) ** 0.5
    b = 0.1 * _rand(M, seed=3)
    y = hiputil.conv_split_tm_hip(x, w, b, pad=(k - 1) // 2, in_slope=slope)
    xin = x.double() if slope == 1.0 else F.leaky_relu(x.double(), slope)
    _check(y, F.conv1d(xin, w.double(), b.double(), padding=(k - 1) // 2))


def _polyphase(wt, u, pad):
    """ConvTranspose1d weight [C_in][C_out][2u] -> the three-tap conv image [u * C_out][C_in][3] (conv_split_tm.hip's header)."""
    Cin, Cout, K = wt.shape
    w = torch.zeros(u * Cout, Cin, 3)
    for r in range(u):
        for j in range(3):
            kk = u * (1 - j) + r + pad
            if 0 <= kk < K:
                w[r * Cout:(r + 1) * Cout, :, j] = wt[:, :, kk].t()
    return w


@pytest.mark.parametrize("B,Cin,Cout,T,u", [(2, 512, 256, 100, 5), (1, 128, 64, 777, 4), (3, 64, 32, 1500, 2), (1, 32, 16, 999, 2), (1, 80, 256, 60, 5)])
def test_conv_transpose_as_three_tap_conv(cuda, lib, B, Cin, Cout, T, u):
    """ups[i] = ConvTranspose1d(C, C/2, 2u, u, padding=u//2+u%2, output_padding=u%2) (models.py:120-123): the [T][u C'] result of the
    polyphase conv, read as [u T][C'], is the transposed conv's output."""
    K, pad = 2 * u, u // 2 + u % 2
    x = _rand(B, Cin, T, seed=1)
    wt = _rand(Cin, Cout, K, seed=2) / (Cin * 2) ** 0.5
    b = 0.1 * _rand(Cout, seed=3)
    y = hiputil.conv_split_tm_hip(x, _polyphase(wt, u, pad), b.repeat(u), pad=1, in_slope=0.1)  # [B, u*Cout, T]
    y = y.view(B, u, Cout, T).permute(0, 2, 3, 1).reshape(B, Cout, T * u)
    ref = F.conv_transpose1d(F.leaky_relu(x.double(), 0.1), wt.double(), b.double(), stride=u, padding=pad, output_padding=u % 2)
    assert ref.shape[-1] == T * u
    _check(y, ref)


def test_residual_tanh_accumulate_and_ragged(cuda, lib):
    B, Cin, M, T, k = 3, 64, 128, 900, 3
    x = _rand(B, Cin, T, seed=1)
    w = _rand(M, Cin, k, seed=2) / (Cin * k) ** 0.5
    b = 0.1 * _rand(M, seed=3)
    res, acc = _rand(B, M, T, seed=4), _rand(B, M, T, seed=5)
    conv = F.conv1d(x.double(), w.double(), b.double(), padding=1)
    _check(hiputil.conv_split_tm_hip(x, w, b, pad=1, res=res), conv + res.double())
    _check(hiputil.conv_split_tm_hip(x, w, b, pad=1, out_act=2), torch.tanh(conv))
    _check(hiputil.conv_split_tm_hip(x, w, b, pad=1, res=res, out_scale=0.5, accumulate_into=acc), acc.double() + 0.5 * (conv + res.double()))
    valid, mul = [100, 33, 1], 9
    y = hiputil.conv_split_tm_hip(x, w, b, pad=1, valid=valid, valid_mul=mul)
    for i, v in enumerate(valid):
        n = v * mul
        xm = x[i:i + 1].double().clone()
        xm[:, :, n:] = 0
        ref = F.conv1d(xm, w.double(), b.double(), padding=1)
        assert float((y[i, :, :n].double() - ref[0, :, :n]).abs().max()) <= 1e-5 * max(1.0, float(ref.pow(2).mean().sqrt()))
        assert bool(torch.isnan(y[i, :, n:]).all())


def test_epilogues_of_the_cbhg(cuda, lib):
    """ReLU -> BatchNorm affine (common/batch_norm_conv.py:11-14), sigmoid, and the highway epilogue (common/highway_network.py:12-17)."""
    B, Cin, M, T = 2, 128, 256, 300
    x = _rand(B, Cin, T, seed=1)
    w = _rand(M, Cin, 3, seed=2) / (Cin * 3) ** 0.5
    b = 0.1 * _rand(M, seed=3)
    ps, pt = 1.0 + 0.2 * _rand(M, seed=4), 0.3 * _rand(M, seed=5)
    conv = F.conv1d(x.double(), w.double(), b.double(), padding=1)
    _check(hiputil.conv_split_tm_hip(x, w, b, pad=1, out_act=1, post=(ps, pt)), torch.relu(conv) * ps.double()[None, :, None] + pt.double()[None, :, None])
    _check(hiputil.conv_split_tm_hip(x, w, b, pad=1, out_act=3), torch.sigmoid(conv))
    gate, res = torch.sigmoid(_rand(B, M, T, seed=6)), _rand(B, M, T, seed=7)
    _check(hiputil.conv_split_tm_hip(x, w, b, pad=1, out_act=4, gate=gate, res=res), gate.double() * torch.relu(conv) + (1 - gate.double()) * res.double())


def test_split_tensors_between_launches(cuda, lib):
    """d_ysplit: the result also as fp16 hi / scaled-lo rows -- bit for bit the split of the fp32 result; x_split: such a tensor as the
    input, staged by copy -- bit for bit the result of the fp32 input (the same halves reach the matrix pipe)."""
    B, Cin, M, T, k = 2, 256, 512, 333, 3
    x = _rand(B, Cin, T, seed=1)
    w = _rand(M, Cin, k, seed=2) / (Cin * k) ** 0.5
    b = 0.1 * _rand(M, seed=3)
    y, ysp, ytm = hiputil.conv_split_tm_hip(x, w, b, pad=1, want_split=True)
    assert torch.equal(ysp, hiputil.split_tensor(ytm))
    y2 = hiputil.conv_split_tm_hip(x, w, b, pad=1, x_split=True)
    assert torch.equal(y2, y)
    _check(y, F.conv1d(x.double(), w.double(), b.double(), padding=1))


def test_maxpool_and_highway_kernels(cuda, lib):
    """MaxPool1d(2, 1, 1)[:T] (sublayer/cbhg.py:20,61-62) and the highway combine on time-major tensors, fp32 and split outputs."""
    import ctypes as C
    from mockingbird_amd import _lib
    L = _lib.lib()
    B, Cc, T = 3, 96, 50
    x = _rand(B, Cc, T, seed=1)
    xt = hiputil.f32_cm_to_tm(x)
    y = torch.empty_like(xt); ysp = torch.zeros(B, T, 2, Cc, dtype=torch.float16, device="cuda")
    _lib.check(L.mb_maxpool2_tm(xt.data_ptr(), y.data_ptr(), ysp.data_ptr(), B, T, Cc, None), "mb_maxpool2_tm")
    torch.cuda.synchronize()
    ref = F.max_pool1d(x, 2, 1, 1)[:, :, :T]
    assert torch.equal(hiputil.f32_tm_to_cm(y).cpu(), ref) and torch.equal(ysp.cpu(), hiputil.split_tensor(y.cpu()))
    hg = _rand(B * T, 2 * Cc, seed=2).cuda()
    out = torch.empty(B * T, Cc, device="cuda"); osp = torch.zeros(B * T, 2, Cc, dtype=torch.float16, device="cuda")
    xr = xt.reshape(B * T, Cc).contiguous()
    _lib.check(L.mb_highway_tm(hg.data_ptr(), xr.data_ptr(), out.data_ptr(), osp.data_ptr(), B * T, Cc, None), "mb_highway_tm")
    torch.cuda.synchronize()
    h, g = hg[:, :Cc].double().cpu(), torch.sigmoid(hg[:, Cc:].double().cpu())
    refh = g * torch.relu(h) + (1 - g) * xr.double().cpu()
    assert float((out.double().cpu() - refh).abs().max()) <= 1e-6
    assert torch.equal(osp.cpu(), hiputil.split_tensor(out.cpu()[None])[0])


@pytest.mark.parametrize("B,Cin,T,k,dil,slope,act", [(3, 32, 1000, 7, 1, 0.01, 2), (2, 64, 517, 7, 1, 0.01, 2), (2, 16, 300, 3, 2, 1.0, 0),
                                                    (1, 32, 40, 1, 1, 0.1, 0)])
def test_one_channel_conv_is_exact_fp32(cuda, lib, B, Cin, T, k, dil, slope, act):
    """mb_conv_c1_tm (conv_post of the generators: leaky_relu -> Conv1d(C, 1, 7) -> tanh, models/vocoder/hifigan/models.py:146-148) against
    float64; exact fp32 FMAs: the error is fp32 rounding of a k * C term sum."""
    g = torch.Generator().manual_seed(B * 1000 + Cin + T)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(1, Cin, k, generator=g) / (Cin * k) ** 0.5
    bias = 0.25
    xa = F.leaky_relu(x.double(), slope) if slope != 1.0 else x.double()
    ref = F.conv1d(xa, w.double(), torch.tensor([bias], dtype=torch.float64), padding=dil * (k - 1) // 2, dilation=dil)[:, 0]
    if act == 2:
        ref = torch.tanh(ref)
    got = hiputil.conv_c1_tm_hip(x, w, bias, dilation=dil, in_slope=slope, out_act=act)
    assert torch.isfinite(got).all()
    assert (got.double() - ref).abs().max() < 2e-6
    # ragged: rows beyond an item's length read as zeros and are not written
    valid = [max(1, (T // 4) // (b + 1)) for b in range(B)]
    got = hiputil.conv_c1_tm_hip(x, w, bias, dilation=dil, in_slope=slope, out_act=act, valid=valid, valid_mul=4)
    for b in range(B):
        n = min(T, valid[b] * 4)
        xb = xa[b:b + 1, :, :n]
        rb = F.conv1d(xb, w.double(), torch.tensor([bias], dtype=torch.float64), padding=dil * (k - 1) // 2, dilation=dil)[0, 0]
        if act == 2:
            rb = torch.tanh(rb)
        assert (got[b, :n].double() - rb).abs().max() < 2e-6
        assert torch.isnan(got[b, n:]).all()


@pytest.mark.parametrize("B,Cin,M,T,k,split", [(8, 256, 640, 600, 3, False), (5, 512, 1280, 1000, 3, False), (9, 512, 1024, 500, 1, True),
                                               (8, 80, 1024, 600, 9, False), (6, 256, 384, 700, 1, False)])
def test_resident_window_kernel_equals_the_ring_kernel(cuda, lib, monkeypatch, B, Cin, M, T, k, split):
    """Convs to several channel groups over enough rows run conv_split_tm_res_kernel (the window laid down once per position tile,
    every group computed from it); MBHIP_DIAG=ctm_nores keeps them on the per-group ring kernel.  Same sums per output: bit-identical,
    residual / accumulate / tanh epilogue, a ragged batch and a narrower last channel group (384 = 256 + 128) included."""
    x = _rand(B, Cin, T, seed=1)
    w = _rand(M, Cin, k, seed=2) / (Cin * k) ** 0.5
    b = 0.1 * _rand(M, seed=3)
    res = _rand(B, M, T, seed=4)
    acc = _rand(B, M, T, seed=5)
    valid = [T - 11 * (i % 4) for i in range(B)]
    outs = []
    for diag in ("", "ctm_nores"):
        monkeypatch.setenv("MBHIP_DIAG", diag)
        y0 = hiputil.conv_split_tm_hip(x, w, b, pad=(k - 1) // 2, in_slope=1.0 if split else 0.1, x_split=split)
        y1 = hiputil.conv_split_tm_hip(x, w, b, pad=(k - 1) // 2, in_slope=1.0 if split else 0.1, x_split=split, res=res, out_act=2,
                                       out_scale=0.5, accumulate_into=acc, valid=valid)
        outs.append((y0, y1))
    monkeypatch.delenv("MBHIP_DIAG")
    assert torch.equal(outs[0][0], outs[1][0])
    for i in range(B):
        assert torch.equal(outs[0][1][i, :, :valid[i]], outs[1][1][i, :, :valid[i]])
    xin = x.double() if split else F.leaky_relu(x.double(), 0.1)
    _check(outs[0][0], F.conv1d(xin, w.double(), b.double(), padding=(k - 1) // 2))


def test_rejects_bad_shapes(cuda, lib):
    from mockingbird_amd._lib import MbHipError
    with pytest.raises(MbHipError, match="unsupported"):
        hiputil.conv_split_tm_hip(_rand(1, 32, 50), _rand(8, 32, 4), None, pad=1)  # even kernel size
