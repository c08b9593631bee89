"""monotonic_align.maximum_path: HIP kernel vs the C oracle, bit-exact (int32 path, float32 value)."""
import numpy as np
import pytest
import torch

from oracle import maximum_path as omp

pytestmark = pytest.mark.gpu


def _case(b, t_t, t_s, seed, ragged=True):
    rng = np.random.default_rng(seed)
    neg = rng.standard_normal((b, t_t, t_s)).astype(np.float32) * 3
    mask = np.zeros((b, t_t, t_s), np.float32)
    for i in range(b):
        ty = rng.integers(max(1, t_t // 2), t_t + 1) if ragged else t_t
        tx = rng.integers(1, min(t_s, ty) + 1) if ragged else min(t_s, t_t)
        mask[i, :ty, :tx] = 1
    return neg, mask


@pytest.mark.parametrize("b,t_t,t_s,seed", [(1, 1, 1, 0), (3, 17, 5, 1), (4, 200, 60, 2), (2, 1000, 200, 3),
                                            (5, 64, 64, 4)])
def test_maximum_path_bit_exact(cuda, lib, b, t_t, t_s, seed):
    from mockingbird_amd.monotonic_align import maximum_path
    neg, mask = _case(b, t_t, t_s, seed)
    ref_path, _ = omp.maximum_path(neg, mask)
    out = maximum_path(torch.from_numpy(neg).cuda(), torch.from_numpy(mask).cuda())
    assert out.dtype == torch.float32 and out.is_cuda
    assert np.array_equal(out.cpu().numpy().astype(np.int32), ref_path)
    # path properties: one token per frame, monotone, starts at 0 ends at t_x-1
    p = out.cpu().numpy()
    for i in range(b):
        ty, tx = int(mask[i].sum(0)[0]), int(mask[i].sum(1)[0])
        rows = p[i, :ty, :]
        assert (rows.sum(1) == 1).all()
        idx = rows.argmax(1)
        assert idx[0] == 0 and idx[-1] == tx - 1 and (np.diff(idx) >= 0).all() and (np.diff(idx) <= 1).all()


def test_maximum_path_empty(cuda, lib):
    from mockingbird_amd.monotonic_align import maximum_path
    out = maximum_path(torch.zeros(0, 4, 3).cuda(), torch.zeros(0, 4, 3).cuda())
    assert out.shape == (0, 4, 3)


def test_maximum_path_accepts_host_tensors(cuda, lib):
    """The reference wrapper is device-agnostic (monotonic_align/__init__.py:10-19: whatever comes in is copied
    to numpy, the path goes back to the caller's device/dtype): host tensors in -> host tensor out, computed on
    the MI355X, input untouched."""
    from mockingbird_amd.monotonic_align import maximum_path
    neg, mask = _case(3, 40, 11, 7)
    ref_path, _ = omp.maximum_path(neg, mask)
    tn = torch.from_numpy(neg.copy())
    out = maximum_path(tn, torch.from_numpy(mask))
    assert not out.is_cuda and out.dtype == torch.float32
    assert np.array_equal(out.numpy().astype(np.int32), ref_path)
    assert np.array_equal(tn.numpy(), neg)  # core.pyx mutates its private copy only
    outd = maximum_path(torch.from_numpy(neg).double(), torch.from_numpy(mask))
    assert outd.dtype == torch.float64 and np.array_equal(outd.numpy().astype(np.int32), ref_path)


def test_maximum_path_in_vits_forward_shape(cuda, lib):
    """the call site in Vits.forward (models/synthesizer/models/vits.py:469-479) restated --
    neg_cent assembled on the device under no_grad from (z_p, m_p, logs_p), attn_mask from the length masks,
    `attn = monotonic_align.maximum_path(neg_cent, attn_mask.squeeze(1)).unsqueeze(1).detach()` and the
    duration target `w = attn.sum(2)`; checked against the C oracle on the same neg_cent."""
    import math
    from mockingbird_amd.monotonic_align import maximum_path
    g = torch.Generator().manual_seed(3)
    b, d, t_s, t_t = 4, 192, 23, 117
    x_lengths = torch.tensor([23, 17, 9, 1])
    y_lengths = torch.tensor([117, 80, 64, 5])
    z_p = torch.randn(b, d, t_t, generator=g).cuda()
    m_p = torch.randn(b, d, t_s, generator=g).cuda()
    logs_p = (0.3 * torch.randn(b, d, t_s, generator=g)).cuda()
    x_mask = (torch.arange(t_s)[None] < x_lengths[:, None]).float().unsqueeze(1).cuda()  # [b, 1, t_s]
    y_mask = (torch.arange(t_t)[None] < y_lengths[:, None]).float().unsqueeze(1).cuda()  # [b, 1, t_t]
    with torch.no_grad():
        s_p_sq_r = torch.exp(-2 * logs_p)
        neg_cent1 = torch.sum(-0.5 * math.log(2 * math.pi) - logs_p, [1], keepdim=True)
        neg_cent2 = torch.matmul(-0.5 * (z_p ** 2).transpose(1, 2), s_p_sq_r)
        neg_cent3 = torch.matmul(z_p.transpose(1, 2), (m_p * s_p_sq_r))
        neg_cent4 = torch.sum(-0.5 * (m_p ** 2) * s_p_sq_r, [1], keepdim=True)
        neg_cent = neg_cent1 + neg_cent2 + neg_cent3 + neg_cent4
        attn_mask = torch.unsqueeze(x_mask, 2) * torch.unsqueeze(y_mask, -1)
        before = neg_cent.clone()
        attn = maximum_path(neg_cent, attn_mask.squeeze(1)).unsqueeze(1).detach()
    assert attn.shape == (b, 1, t_t, t_s) and attn.is_cuda and attn.dtype == neg_cent.dtype
    assert torch.equal(neg_cent, before)  # the caller's tensor is not the DP scratch
    ref_path, _ = omp.maximum_path(before.cpu().numpy(), attn_mask.squeeze(1).cpu().numpy())
    assert np.array_equal(attn[:, 0].cpu().numpy().astype(np.int32), ref_path)
    w = attn.sum(2)  # [b, 1, t_s] durations: every frame of an utterance is assigned to exactly one token
    assert torch.equal(w.sum(-1)[:, 0].cpu(), y_lengths.float())
    assert bool((w[:, 0].cpu() * (1 - x_mask[:, 0].cpu())).sum() == 0)
