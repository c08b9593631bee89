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
