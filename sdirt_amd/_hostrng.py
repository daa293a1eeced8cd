"""torch.rand on the CPU default generator -- the reference's source of pupil samples
(deeplens/optics.py:483-484) -- through libsdirt_dp.so's block-wise MT19937 (sdirt_host_uniform_fill):
the same numbers, the same generator state afterwards, a fraction of the time (torch produces one
number per ~1.6 ns; 44096 of them per PSFNet fitting batch).

Safety: before its first use the fast path is checked against torch.rand itself on a private
generator (values AND the state left behind, across block boundaries); any difference -- a torch
build with another state layout, say -- switches it off for the process and torch.rand is used.
SDIRT_TORCH_RAND=1 forces torch.rand."""
import ctypes as C
import os

import torch

from . import _lib

_ok = None


def _selfcheck():
    h = _lib.lib()
    g = torch.Generator()
    for seed, pre, n in ((5, 0, 700), (6, 623, 1300), (7, 10, 5000)):
        g.manual_seed(seed)
        if pre:
            torch.rand(pre, generator=g)
        st = g.get_state().clone()
        want = torch.rand(n, generator=g)
        after = g.get_state()
        got = torch.empty(n)
        if h.sdirt_host_uniform_fill(C.c_void_p(st.data_ptr()), st.numel(), n, C.c_void_p(got.data_ptr())) != 0:
            return False
        if not (torch.equal(got, want) and torch.equal(st, after)):
            return False
    return True


def enabled():
    global _ok
    if _ok is None:
        _ok = os.environ.get("SDIRT_TORCH_RAND") != "1" and _selfcheck()
    return _ok


def rand_into(out):
    """out (float32, CPU, contiguous, 1-D) <- torch.rand(out.numel()) from the default CPU generator."""
    if enabled():
        st = torch.get_rng_state()
        if _lib.lib().sdirt_host_uniform_fill(C.c_void_p(st.data_ptr()), st.numel(), out.numel(),
                                              C.c_void_p(out.data_ptr())) == 0:
            torch.set_rng_state(st)
            return out
    return torch.rand(out.numel(), out=out)
