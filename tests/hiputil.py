"""Thin helpers to call the C ABI primitives from tests."""
import ctypes as C

import torch

from mockingbird_amd import _lib


def pack_conv(w: torch.Tensor, transposed=False, up=1, pad=0):
    """torch-layout conv weight -> packed device tensor."""
    L = _lib.lib()
    w = w.detach().float().contiguous().cpu()
    if transposed:
        c_in, c_out, k = w.shape
    else:
        c_out, c_in, k = w.shape
    n = L.mb_conv1d_packed_floats(c_out, c_in, k, up)
    out = torch.empty(n, dtype=torch.float32)
    _lib.check(L.mb_conv1d_pack(w.data_ptr(), c_out, c_in, k, up, int(transposed), pad, out.data_ptr()),
               "mb_conv1d_pack")
    return out, (c_out, c_in, k)


def conv1d_hip(x, w, bias=None, *, dilation=1, pad=0, transposed=False, up=1, in_act=0, in_slope=0.0,
               in_scale=1.0, out_act=0, out_scale=1.0, res=None, accumulate_into=None, post=None,
               transpose_out=False, in_repeat=1, stride=1, out_slope=0.0, device="cuda"):
    """x [B, Cin, T] CPU/GPU tensor -> y (new tensor, or accumulate_into updated in place)."""
    L = _lib.lib()
    packed, (c_out, c_in, k) = pack_conv(w, transposed, up, pad)
    dev = torch.device(device)
    x = x.float().contiguous().to(dev)
    B, _, t_src = x.shape
    t_in = t_src * in_repeat
    t_out = t_in * up if transposed else (t_in + 2 * pad - dilation * (k - 1) - 1) // stride + 1
    a = _lib.ConvArgs()
    pw = packed.to(dev)
    pb = bias.float().contiguous().to(dev) if bias is not None else None
    pres = res.float().contiguous().to(dev) if res is not None else None
    if accumulate_into is not None:
        y = accumulate_into.float().contiguous().to(dev)
    elif transpose_out:
        y = torch.full((B, t_out, c_out), float("nan"), device=dev)
    else:
        y = torch.full((B, c_out, t_out), float("nan"), device=dev)
    ps = pt = None
    if post is not None:
        ps, pt = post[0].float().contiguous().to(dev), post[1].float().contiguous().to(dev)
    a.d_x, a.d_wpacked, a.d_bias, a.d_res = x.data_ptr(), pw.data_ptr(), (pb.data_ptr() if pb is not None else None), (pres.data_ptr() if pres is not None else None)
    a.d_post_scale, a.d_post_shift = (ps.data_ptr() if ps is not None else None), (pt.data_ptr() if pt is not None else None)
    a.d_y = y.data_ptr()
    a.x_bstride, a.y_bstride, a.res_bstride = c_in * t_src, c_out * t_out, c_out * t_out
    a.batch, a.c_in, a.c_out, a.t_in, a.t_out = B, c_in, c_out, t_in, t_out
    a.ksize, a.dilation, a.pad, a.up = k, dilation, pad, (up if transposed else 1)
    a.in_act, a.in_slope, a.in_scale = in_act, in_slope, in_scale
    a.out_act, a.out_scale, a.accumulate = out_act, out_scale, int(accumulate_into is not None)
    a.in_repeat, a.transpose_out = in_repeat, int(transpose_out)
    a.down, a.out_slope = stride, out_slope
    _lib.check(L.mb_conv1d(C.byref(a), _lib.stream_ptr()), "mb_conv1d")
    torch.cuda.synchronize()
    return y


def conv1d_f16_hip(x, w, bias=None, *, dilation=1, pad=0, transposed=False, up=1, in_act=0, in_slope=0.0,
                   out_act=0, out_scale=1.0, res=None, accumulate_into=None, in_repeat=1, y_f32=False,
                   device="cuda"):
    """fp16 primitive (mb_conv1d_f16).  x / res / accumulate_into are [B, C, T] float tensors in the
    reference's layout; they are converted to time-major fp16 with the ABI's own converter and the
    result comes back as [B, Cout, T_out] float32."""
    L = _lib.lib()
    dev = torch.device(device)
    w = w.detach().float().contiguous().cpu()
    if transposed:
        c_in, c_out, k = w.shape
    else:
        c_out, c_in, k = w.shape
    nh = L.mb_conv1d_f16_packed_halves(c_out, c_in, k, up)
    packed = torch.empty(nh, dtype=torch.float16)
    _lib.check(L.mb_conv1d_f16_pack(w.data_ptr(), c_out, c_in, k, up, int(transposed), pad, packed.data_ptr()),
               "mb_conv1d_f16_pack")
    pw = packed.to(dev)

    def to_tm(t):
        t = t.float().contiguous().to(dev)
        B, Cc, Tt = t.shape
        out = torch.empty(B, Tt, Cc, dtype=torch.float16, device=dev)
        _lib.check(L.mb_f32_to_f16_tm(t.data_ptr(), out.data_ptr(), B, Cc, Tt, _lib.stream_ptr()), "mb_f32_to_f16_tm")
        return out

    xh = to_tm(x)
    B, t_src, _ = xh.shape
    t_in = t_src * in_repeat
    t_out = t_in * up if transposed else t_in + 2 * pad - dilation * (k - 1)
    pb = bias.float().contiguous().to(dev) if bias is not None else None
    pres = to_tm(res) if res is not None else None
    if accumulate_into is not None:
        y = accumulate_into.float().contiguous().to(dev).transpose(1, 2).contiguous() if y_f32 else to_tm(accumulate_into)
    elif y_f32:
        y = torch.full((B, t_out, c_out), float("nan"), device=dev)
    else:
        y = torch.full((B, t_out, c_out), float("nan"), dtype=torch.float16, device=dev)
    a = _lib.ConvF16Args()
    a.d_x, a.d_wpacked = xh.data_ptr(), pw.data_ptr()
    a.d_bias = pb.data_ptr() if pb is not None else None
    a.d_res = pres.data_ptr() if pres is not None else None
    a.d_y = y.data_ptr()
    a.x_bstride, a.y_bstride, a.res_bstride = c_in * t_src, c_out * t_out, c_out * t_out
    a.batch, a.c_in, a.c_out, a.t_in, a.t_out = B, c_in, c_out, t_in, t_out
    a.ksize, a.dilation, a.pad, a.up = k, dilation, pad, (up if transposed else 1)
    a.in_act, a.in_slope = in_act, in_slope
    a.out_act, a.out_scale, a.accumulate = out_act, out_scale, int(accumulate_into is not None)
    a.in_repeat, a.y_f32 = in_repeat, int(y_f32)
    _lib.check(L.mb_conv1d_f16(C.byref(a), _lib.stream_ptr()), "mb_conv1d_f16")
    torch.cuda.synchronize()
    if y_f32:
        return y.transpose(1, 2).contiguous()
    out = torch.empty(B, c_out, t_out, device=dev)
    _lib.check(L.mb_f16_tm_to_f32(y.data_ptr(), out.data_ptr(), B, c_out, t_out, _lib.stream_ptr()), "mb_f16_tm_to_f32")
    torch.cuda.synchronize()
    return out


def relerr(a: torch.Tensor, b: torch.Tensor):
    a, b = a.double().cpu(), b.double().cpu()
    d = (a - b).abs()
    return dict(max_abs=float(d.max()), rms=float(d.pow(2).mean().sqrt()),
                rel_rms=float(d.pow(2).mean().sqrt() / (b.pow(2).mean().sqrt() + 1e-30)),
                ref_rms=float(b.pow(2).mean().sqrt()), nan=int(torch.isnan(a).sum()))


def resblock_pair_f16_hip(x, w1, b1, w2, b2, *, dilation=1, slope=0.1, out_scale=1.0, accumulate_into=None,
                          device="cuda"):
    """Fused ResBlock unit (mb_resblock_pair_f16).  x / accumulate_into are [B, C, T] float tensors
    (reference layout); returns [B, C, T] float32."""
    L = _lib.lib()
    dev = torch.device(device)
    w1 = w1.detach().float().contiguous().cpu()
    w2 = w2.detach().float().contiguous().cpu()
    Cc, _, k = w1.shape
    packed = torch.empty(L.mb_resblock_pair_f16_packed_halves(Cc, k), dtype=torch.float16)
    _lib.check(L.mb_resblock_pair_f16_pack(w1.data_ptr(), w2.data_ptr(), Cc, k, packed.data_ptr()),
               "mb_resblock_pair_f16_pack")
    pw = packed.to(dev)

    def to_tm(t):
        t = t.float().contiguous().to(dev)
        B, C_, Tt = t.shape
        out = torch.empty(B, Tt, C_, dtype=torch.float16, device=dev)
        _lib.check(L.mb_f32_to_f16_tm(t.data_ptr(), out.data_ptr(), B, C_, Tt, _lib.stream_ptr()), "mb_f32_to_f16_tm")
        return out

    xh = to_tm(x)
    B, T, _ = xh.shape
    y = to_tm(accumulate_into) if accumulate_into is not None else \
        torch.full((B, T, Cc), float("nan"), dtype=torch.float16, device=dev)
    pb1, pb2 = b1.float().contiguous().to(dev), b2.float().contiguous().to(dev)
    a = _lib.ResPairF16Args()
    a.d_x, a.d_y, a.d_wpacked, a.d_b1, a.d_b2 = xh.data_ptr(), y.data_ptr(), pw.data_ptr(), pb1.data_ptr(), pb2.data_ptr()
    a.batch, a.channels, a.t, a.ksize, a.dilation = B, Cc, T, k, dilation
    a.slope, a.out_scale, a.accumulate = slope, out_scale, int(accumulate_into is not None)
    _lib.check(L.mb_resblock_pair_f16(C.byref(a), _lib.stream_ptr()), "mb_resblock_pair_f16")
    torch.cuda.synchronize()
    return y.float().transpose(1, 2).contiguous().cpu()


def f32_cm_to_tm(t, device="cuda"):
    """[B, C, T] float -> device fp32 [B, T, C] through mb_f32_cm_to_tm."""
    L = _lib.lib()
    t = t.float().contiguous().to(torch.device(device))
    B, C_, Tt = t.shape
    out = torch.empty(B, Tt, C_, dtype=torch.float32, device=t.device)
    _lib.check(L.mb_f32_cm_to_tm(t.data_ptr(), out.data_ptr(), B, C_, Tt, _lib.stream_ptr()), "mb_f32_cm_to_tm")
    return out


def f32_tm_to_cm(t):
    """device fp32 [B, T, C] -> device fp32 [B, C, T] through mb_f32_tm_to_cm."""
    L = _lib.lib()
    B, Tt, C_ = t.shape
    out = torch.empty(B, C_, Tt, dtype=torch.float32, device=t.device)
    _lib.check(L.mb_f32_tm_to_cm(t.data_ptr(), out.data_ptr(), B, C_, Tt, _lib.stream_ptr()), "mb_f32_tm_to_cm")
    return out


def pack_pair_split(w1, w2, device="cuda"):
    """(packed fp16 image on the device, unscale1, unscale2) of mb_resblock_pair_split_pack."""
    L = _lib.lib()
    w1 = w1.detach().float().contiguous().cpu()
    w2 = w2.detach().float().contiguous().cpu()
    Cc, _, k = w1.shape
    packed = torch.empty(L.mb_resblock_pair_split_packed_halves(Cc, k), dtype=torch.float16)
    us = torch.zeros(2, dtype=torch.float32)
    _lib.check(L.mb_resblock_pair_split_pack(w1.data_ptr(), w2.data_ptr(), Cc, k, packed.data_ptr(), us.data_ptr()),
               "mb_resblock_pair_split_pack")
    return packed.to(torch.device(device)), float(us[0]), float(us[1])


def resblock_pair_split_hip(x, w1, b1, w2, b2, *, dilation=1, slope=0.1, out_scale=1.0, accumulate_into=None, valid=None,
                            valid_mul=1, device="cuda"):
    """Fused ResBlock unit at fp32-grade precision (mb_resblock_pair_split).  x / accumulate_into are [B, C, T] float tensors
    (reference layout, turned time-major on the device); returns [B, C, T] float32 (rows beyond `valid` keep their NaN fill)."""
    L = _lib.lib()
    dev = torch.device(device)
    Cc, _, k = w1.shape
    if not L.mb_resblock_pair_split_supported(Cc, k, dilation):
        raise _lib.MbHipError("mb_resblock_pair_split: unsupported shape")
    pw, us1, us2 = pack_pair_split(w1, w2, device)
    xt = f32_cm_to_tm(x, device)
    B, T, _ = xt.shape
    y = f32_cm_to_tm(accumulate_into, device) if accumulate_into is not None else \
        torch.full((B, T, Cc), float("nan"), dtype=torch.float32, device=dev)
    pb1, pb2 = b1.float().contiguous().to(dev), b2.float().contiguous().to(dev)
    vt = torch.tensor(valid, dtype=torch.int32, device=dev) if valid is not None else None
    a = _lib.ResPairSplitArgs()
    a.d_x, a.d_y, a.d_wpacked, a.d_b1, a.d_b2 = xt.data_ptr(), y.data_ptr(), pw.data_ptr(), pb1.data_ptr(), pb2.data_ptr()
    a.batch, a.channels, a.t, a.ksize, a.dilation = B, Cc, T, k, dilation
    a.slope, a.out_scale, a.unscale1, a.unscale2 = slope, out_scale, us1, us2
    a.accumulate = int(accumulate_into is not None)
    a.d_valid, a.valid_mul = (vt.data_ptr() if vt is not None else None), valid_mul
    _lib.check(L.mb_resblock_pair_split(C.byref(a), _lib.stream_ptr()), "mb_resblock_pair_split")
    torch.cuda.synchronize()
    return f32_tm_to_cm(y).cpu()


def split_tensor(x_tm):
    """fp32 [B, T, C] -> the split tensor fp16 [B, T, 2, C] of the error-compensated paths: hi = fp16(x), lo = fp16((x - hi) * 2^11)
    (values clamped to fp16's range first: common.h split_pair)."""
    v = x_tm.float().clamp(-65504.0, 65504.0)
    hi = v.half()
    lo = ((v - hi.float()) * 2048.0).half()
    return torch.stack((hi, lo), dim=2).contiguous()


def conv_split_tm_hip(x, w, bias=None, *, pad=0, dilation=1, in_slope=1.0, res=None, out_act=0, out_scale=1.0, accumulate_into=None,
                      valid=None, valid_mul=1, device="cuda", x_split=False, want_split=False, gate=None, post=None):
    """Time-major split conv (mb_conv_split_tm).  x [B, C_in, T] float (reference layout, turned on the device), w [M, C_in, k] (torch
    Conv1d layout), res / accumulate_into [B, M, T]; returns [B, M, T] float32 (rows beyond `valid` keep their NaN fill)."""
    L = _lib.lib()
    dev = torch.device(device)
    w = w.detach().float().contiguous().cpu()
    M, Cin, k = w.shape
    if not L.mb_conv_split_tm_supported(M, Cin, k, dilation):
        raise _lib.MbHipError("mb_conv_split_tm: unsupported shape")
    packed = torch.empty(L.mb_conv_split_tm_packed_halves(M, Cin, k), dtype=torch.float16)
    us = torch.zeros(1, dtype=torch.float32)
    _lib.check(L.mb_conv_split_tm_pack(w.data_ptr(), M, Cin, k, packed.data_ptr(), us.data_ptr()), "mb_conv_split_tm_pack")
    pw = packed.to(dev)
    xt = f32_cm_to_tm(x, device)
    B, T, _ = xt.shape
    if x_split:  # the same values handed over as a split tensor (staged by copy)
        xt = split_tensor(xt)
    y = f32_cm_to_tm(accumulate_into, device) if accumulate_into is not None else \
        torch.full((B, T, M), float("nan"), dtype=torch.float32, device=dev)
    rt = f32_cm_to_tm(res, device) if res is not None else None
    gt = f32_cm_to_tm(gate, device) if gate is not None else None
    ps, pt = (post[0].float().contiguous().to(dev), post[1].float().contiguous().to(dev)) if post is not None else (None, None)
    ysp = torch.zeros(B, T, 2, M, dtype=torch.float16, device=dev) if want_split else None
    pb = bias.float().contiguous().to(dev) if bias is not None else None
    vt = torch.tensor(valid, dtype=torch.int32, device=dev) if valid is not None else None
    a = _lib.ConvSplitTmArgs()
    a.d_x, a.d_y, a.d_wpacked = xt.data_ptr(), y.data_ptr(), pw.data_ptr()
    a.d_bias = pb.data_ptr() if pb is not None else None
    a.d_res = rt.data_ptr() if rt is not None else None
    a.batch, a.t, a.c_in, a.c_out, a.ksize, a.dilation, a.pad = B, T, Cin, M, k, dilation, pad
    a.in_slope, a.unscale, a.out_scale, a.out_act = in_slope, float(us[0]), out_scale, out_act
    a.accumulate = int(accumulate_into is not None)
    a.d_valid, a.valid_mul = (vt.data_ptr() if vt is not None else None), valid_mul
    a.d_gate = gt.data_ptr() if gt is not None else None
    a.d_post_scale, a.d_post_shift = (ps.data_ptr(), pt.data_ptr()) if ps is not None else (None, None)
    a.x_split = int(x_split)
    a.d_ysplit = ysp.data_ptr() if ysp is not None else None
    _lib.check(L.mb_conv_split_tm(C.byref(a), _lib.stream_ptr()), "mb_conv_split_tm")
    torch.cuda.synchronize()
    out = f32_tm_to_cm(y).cpu()
    return (out, ysp.cpu(), y.cpu()) if want_split else out


def conv_c1_tm_hip(x, w, bias=0.0, *, dilation=1, in_slope=1.0, out_act=0, valid=None, valid_mul=1, device="cuda"):
    """One-output-channel conv on a time-major tensor (mb_conv_c1_tm, the generators' conv_post).  x [B, C, T], w [1, C, k] (torch
    layout); returns [B, T] float32 (rows beyond `valid` keep their NaN fill)."""
    L = _lib.lib()
    dev = torch.device(device)
    _, Cin, k = w.shape
    w1 = w.detach().float()[0].t().contiguous().to(dev)  # [k][C]
    xt = f32_cm_to_tm(x, device)
    B, T, _ = xt.shape
    y = torch.full((B, T), float("nan"), dtype=torch.float32, device=dev)
    vt = torch.tensor(valid, dtype=torch.int32, device=dev) if valid is not None else None
    _lib.check(L.mb_conv_c1_tm(xt.data_ptr(), w1.data_ptr(), float(bias), y.data_ptr(), B, T, Cin, k, dilation, dilation * (k - 1) // 2,
                               in_slope, out_act, vt.data_ptr() if vt is not None else None, valid_mul, _lib.stream_ptr()), "mb_conv_c1_tm")
    torch.cuda.synchronize()
    return y.cpu()


def resblock_stage_f16_hip(x, chains, *, slope=0.1, out_scale=0.0, valid=None, valid_mul=1, accumulate_into=None, device="cuda"):
    """One stage's ResBlock group in one launch (mb_resblock_stage_f16).  x: [B, C, T] float; chains: list (one per ResBlock)
    of lists (one per unit) of (w1, b1, w2, b2, dilation); returns [B, C, T] float32 (rows beyond `valid` are left NaN)."""
    L = _lib.lib()
    dev = torch.device(device)
    nk, nd = len(chains), len(chains[0])
    Cc = chains[0][0][0].shape[0]
    ks = (C.c_int * nk)(*[ch[0][0].shape[-1] for ch in chains])
    dil = (C.c_int * (nk * nd))(*[u[4] for ch in chains for u in ch])
    if not L.mb_resblock_stage_f16_supported(Cc, nk, ks, nd, dil):
        raise _lib.MbHipError("mb_resblock_stage_f16: unsupported shape")
    w1 = [u[0].detach().float().contiguous().cpu() for ch in chains for u in ch]
    w2 = [u[2].detach().float().contiguous().cpu() for ch in chains for u in ch]
    p1 = (C.c_void_p * len(w1))(*[w.data_ptr() for w in w1])
    p2 = (C.c_void_p * len(w2))(*[w.data_ptr() for w in w2])
    packed = torch.empty(L.mb_resblock_stage_f16_packed_halves(Cc, nk, ks, nd), dtype=torch.float16)
    _lib.check(L.mb_resblock_stage_f16_pack(p1, p2, Cc, nk, ks, nd, packed.data_ptr()), "mb_resblock_stage_f16_pack")
    pw = packed.to(dev)
    bias = torch.stack([torch.stack([u[1].float(), u[3].float()]) for ch in chains for u in ch]).contiguous().to(dev)
    xt = x.float().contiguous().to(dev)
    B, _, T = xt.shape
    xh = torch.empty(B, T, Cc, dtype=torch.float16, device=dev)
    _lib.check(L.mb_f32_to_f16_tm(xt.data_ptr(), xh.data_ptr(), B, Cc, T, _lib.stream_ptr()), "mb_f32_to_f16_tm")
    if accumulate_into is not None:
        at = accumulate_into.float().contiguous().to(dev)
        y = torch.empty(B, T, Cc, dtype=torch.float16, device=dev)
        _lib.check(L.mb_f32_to_f16_tm(at.data_ptr(), y.data_ptr(), B, Cc, T, _lib.stream_ptr()), "mb_f32_to_f16_tm")
    else:
        y = torch.full((B, T, Cc), float("nan"), dtype=torch.float16, device=dev)
    a = _lib.ResStageF16Args()
    a.accumulate = int(accumulate_into is not None)
    a.d_x, a.d_y, a.d_wpacked, a.d_bias = xh.data_ptr(), y.data_ptr(), pw.data_ptr(), bias.data_ptr()
    a.batch, a.channels, a.t, a.num_kernels, a.num_dilations = B, Cc, T, nk, nd
    for j, ch in enumerate(chains):
        a.ksize[j] = ch[0][0].shape[-1]
        for u, unit in enumerate(ch):
            a.dilation[j][u] = unit[4]
    a.slope, a.out_scale = slope, out_scale
    vd = None
    if valid is not None:
        vd = torch.tensor(valid, dtype=torch.int32, device=dev)
        a.d_valid, a.valid_mul = vd.data_ptr(), valid_mul
    _lib.check(L.mb_resblock_stage_f16(C.byref(a), _lib.stream_ptr()), "mb_resblock_stage_f16")
    torch.cuda.synchronize()
    return y.float().transpose(1, 2).contiguous().cpu()


def resblock_stage_f32_hip(x, chains, *, slope=0.1, out_scale=0.0, valid=None, valid_mul=1, accumulate_into=None, device="cuda"):
    """One stage's ResBlock group (or one ResBlock / one unit of it) in one launch at the reference's precision (mb_resblock_stage_f32:
    fp32 [B, C, T] in and out).  chains: list (one per ResBlock) of lists (one per unit) of (w1, b1, w2, b2, dilation); returns
    [B, C, T] float32 (positions beyond `valid` are left NaN)."""
    L = _lib.lib()
    dev = torch.device(device)
    nk, nd = len(chains), len(chains[0])
    Cc = chains[0][0][0].shape[0]
    ks = (C.c_int * nk)(*[ch[0][0].shape[-1] for ch in chains])
    dil = (C.c_int * (nk * nd))(*[u[4] for ch in chains for u in ch])
    if not L.mb_resblock_stage_f32_supported(Cc, nk, ks, nd, dil):
        raise _lib.MbHipError("mb_resblock_stage_f32: unsupported shape")
    w1 = [u[0].detach().float().contiguous().cpu() for ch in chains for u in ch]
    w2 = [u[2].detach().float().contiguous().cpu() for ch in chains for u in ch]
    p1 = (C.c_void_p * len(w1))(*[w.data_ptr() for w in w1])
    p2 = (C.c_void_p * len(w2))(*[w.data_ptr() for w in w2])
    packed = torch.empty(L.mb_resblock_stage_f32_packed_halves(Cc, nk, ks, nd), dtype=torch.float16)
    unscale = torch.empty(nk * nd * 2, dtype=torch.float32)
    _lib.check(L.mb_resblock_stage_f32_pack(p1, p2, Cc, nk, ks, nd, packed.data_ptr(), unscale.data_ptr()), "mb_resblock_stage_f32_pack")
    pw = packed.to(dev)
    bias = torch.cat([torch.stack([torch.stack([u[1].float(), u[3].float()]) for ch in chains for u in ch]).flatten(), unscale]).contiguous().to(dev)
    xt = x.float().contiguous().to(dev)
    B, _, T = xt.shape
    y = accumulate_into.float().contiguous().to(dev) if accumulate_into is not None else torch.full((B, Cc, T), float("nan"), device=dev)
    a = _lib.ResStageF16Args()
    a.accumulate = int(accumulate_into is not None)
    a.d_x, a.d_y, a.d_wpacked, a.d_bias = xt.data_ptr(), y.data_ptr(), pw.data_ptr(), bias.data_ptr()
    a.batch, a.channels, a.t, a.num_kernels, a.num_dilations = B, Cc, T, nk, nd
    for j, ch in enumerate(chains):
        a.ksize[j] = ch[0][0].shape[-1]
        for u, unit in enumerate(ch):
            a.dilation[j][u] = unit[4]
    a.slope, a.out_scale = slope, out_scale
    vd = None
    if valid is not None:
        vd = torch.tensor(valid, dtype=torch.int32, device=dev)
        a.d_valid, a.valid_mul = vd.data_ptr(), valid_mul
    _lib.check(L.mb_resblock_stage_f32(C.byref(a), _lib.stream_ptr()), "mb_resblock_stage_f32")
    torch.cuda.synchronize()
    return y.cpu()
