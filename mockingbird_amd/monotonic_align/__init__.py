"""Drop-in for monotonic_align/__init__.py:6-19 -- same ``maximum_path(neg_cent, mask)``
signature (torch in, torch out, same device/dtype) -- with the DP of
monotonic_align/core.pyx:7-42 running on the MI355X (mb_maximum_path) instead of
a device->host->device round trip through Cython."""
import torch

from .. import _lib


def maximum_path(neg_cent, mask):
    """neg_cent: [b, t_t, t_s]; mask: [b, t_t, t_s]."""
    device, dtype = neg_cent.device, neg_cent.dtype
    if not neg_cent.is_cuda:
        raise _lib.MbHipError("maximum_path needs CUDA(HIP) tensors; there is no CPU path")
    values = neg_cent.detach().to(torch.float32).contiguous().clone()   # the kernel mutates `value` in place
    path = torch.zeros(values.shape, dtype=torch.int32, device=device)  # __init__.py:13
    t_t_max = mask.sum(1)[:, 0].to(torch.int32).contiguous()            # __init__.py:15
    t_s_max = mask.sum(2)[:, 0].to(torch.int32).contiguous()            # __init__.py:16
    b, t_t, t_s = values.shape
    _lib.check(_lib.lib().mb_maximum_path(_lib.ptr(path), _lib.ptr(values), _lib.ptr(t_t_max),
                                          _lib.ptr(t_s_max), b, t_t, t_s, _lib.stream_ptr()),
               "mb_maximum_path")
    return path.to(device=device, dtype=dtype)
