"""Image-space per-pixel dual-pixel PSF convolution (consumer of the PSFs).

Same signatures as deeplens/render_psf.py:76-188.  The reference materialises
an unfold buffer [B, C*ks*ks, H*W] (1 GB at 512x768, ks 21) and multiplies it
with the per-pixel kernels; here one HIP kernel gathers the ks*ks neighbourhood
of every output pixel and reads each per-pixel kernel exactly once.
"""
import torch

from . import _lib
from .basics import dptr, stream_ptr


def _render(input, psf, kernel_size, half):
    if input.dim() < 4:
        input = input.unsqueeze(0)
    if input.device.type != "cuda":
        raise _lib.SdirtError("sdirt_amd renders on the GPU only (no CPU fallback)")
    b, c, h, w = input.shape
    img = input.to(torch.float32).contiguous()
    k = psf.to(torch.float32).reshape(b, h, w, 2, kernel_size, kernel_size).contiguous()
    rl = torch.empty((b, c, h, w), dtype=torch.float32, device=input.device)
    rr = torch.empty_like(rl)
    _lib.check(_lib.lib().sdirt_local_psf_render(dptr(img), dptr(k), b, c, h, w, kernel_size,
                                                 1 if half else 0, dptr(rl), dptr(rr),
                                                 stream_ptr(input.device)))
    return rl.to(input.dtype), rr.to(input.dtype)


def local_psf_render(input, psf, kernel_size=11, val=False):
    """render_psf.py:76-118 (fp16 arithmetic) -> (rl, rr) [N,C,H,W]."""
    return _render(input, psf, kernel_size, half=True)


def local_psf_render_fast(input, psf, kernel_size=11, val=False):
    """render_psf.py:120-155 (fp16 arithmetic) -> (rl, rr) [N,C,H,W]."""
    return _render(input, psf, kernel_size, half=True)


def local_dp_psf_render(input, dp_psf, kernel_size=21):
    """render_psf.py:157-188 (fp32) -> [N, 2C, H, W] = cat(left, right)."""
    rl, rr = _render(input, dp_psf, kernel_size, half=False)
    return torch.cat([rl, rr], dim=1)
