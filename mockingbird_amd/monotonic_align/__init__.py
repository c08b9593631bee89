"""Drop-in for monotonic_align/__init__.py:6-19 -- same ``maximum_path(neg_cent, mask)``
signature (torch in, torch out, same device/dtype) -- with the DP of
monotonic_align/core.pyx:7-42 running on the MI355X (mb_maximum_path) instead of
a device->host->device round trip through Cython.

The reference wrapper is device-agnostic (it copies whatever it gets to numpy, :12-14); so is this one:
tensors that live on the host are copied to the GPU, the DP runs there, and the path comes back on the
caller's device.  There is no host implementation -- without a GPU the call fails loudly."""
import torch

from .. import _lib


def maximum_path(neg_cent, mask):
    """neg_cent: [b, t_t, t_s]; mask: [b, t_t, t_s]."""
    device, dtype = neg_cent.device, neg_cent.dtype
    if not neg_cent.is_cuda:
        if not torch.cuda.is_available():
            raise _lib.MbHipError("maximum_path: no MI355X visible; this build has no CPU path")
        run_dev = torch.device("cuda", torch.cuda.current_device())
    else:
        run_dev = device
    values = neg_cent.detach().to(run_dev, torch.float32).contiguous()
    if values.data_ptr() == neg_cent.data_ptr():
        values = values.clone()                                            # the kernel mutates `value` in place
    mask = mask.detach().to(run_dev)
    path = torch.zeros(values.shape, dtype=torch.int32, device=run_dev)    # __init__.py:13
    t_t_max = mask.sum(1)[:, 0].to(torch.int32).contiguous()               # __init__.py:15
    t_s_max = mask.sum(2)[:, 0].to(torch.int32).contiguous()               # __init__.py:16
    b, t_t, t_s = values.shape
    with torch.cuda.device(run_dev):
        _lib.check(_lib.lib().mb_maximum_path(_lib.ptr(path), _lib.ptr(values), _lib.ptr(t_t_max),
                                              _lib.ptr(t_s_max), b, t_t, t_s, _lib.stream_ptr()),
                   "mb_maximum_path")
    return path.to(device=device, dtype=dtype)
