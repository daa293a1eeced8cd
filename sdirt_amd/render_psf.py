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


def psfnet_render(input, raw_l, raw_r, kernel_size):
    """PSFNet.pred + local_psf_render_fast (psfnet.py:317-336, 702-707; render_psf.py:120-155)
    in one pass over the network's raw fp16 outputs raw_l = net(x, y, z), raw_r = net(-x, y, z),
    each [B,H,W,ks,ks]: the stacked, flipped, normalised per-pixel kernels are formed in LDS
    only.  -> (rl, rr) [B,C,H,W] fp32 holding fp16 values, as local_psf_render_fast returns."""
    if input.device.type != "cuda":
        raise _lib.SdirtError("sdirt_amd renders on the GPU only (no CPU fallback)")
    b, c, h, w = input.shape
    img = input.to(torch.float32).contiguous()
    raw_l = raw_l.to(torch.float16).reshape(b, h, w, kernel_size * kernel_size).contiguous()
    raw_r = raw_r.to(torch.float16).reshape(b, h, w, kernel_size * kernel_size).contiguous()
    if raw_l.data_ptr() % 16:                  # the kernel stages with 16-byte loads
        raw_l = raw_l.clone()
    if raw_r.data_ptr() % 16:
        raw_r = raw_r.clone()
    rl = torch.empty((b, c, h, w), dtype=torch.float32, device=input.device)
    rr = torch.empty_like(rl)
    _lib.check(_lib.lib().sdirt_psfnet_render(dptr(img), dptr(raw_l), dptr(raw_r), b, c, h, w,
                                              kernel_size, dptr(rl), dptr(rr),
                                              stream_ptr(input.device)))
    return rl, rr


def render_psf(img, psf):
    """render_psf.py:12-28: one PSF for the whole image, [B,C,H,W] x [C,ks,ks].  A plain
    grouped convolution with reflect padding: dense conv work that stock PyTorch-ROCm
    (MIOpen) already covers -- kept for API completeness, not a custom kernel."""
    _, ks, _ = psf.shape
    padding = int(ks / 2)
    k = torch.flip(psf, [1, 2]).unsqueeze(1)
    img_pad = torch.nn.functional.pad(img, (padding, padding, padding, padding), mode="reflect")
    return torch.nn.functional.conv2d(img_pad, k, groups=img.shape[1], padding=0, bias=None)


def render_psf_map(img, psf_map, grid):
    """render_psf.py:31-73: a grid x grid mosaic of PSFs, one per image patch."""
    assert img.dim() == 4, "Input image should be [B, C, H, W]"
    Cpsf, Hpsf, Wpsf = psf_map.shape
    assert Hpsf % grid == 0 and Wpsf % grid == 0, "PSF map size should be divisible by grid"
    ks = int(Hpsf / grid)
    assert ks % 2 == 1, "PSF kernel size should be odd"
    B, C, H, W = img.shape
    assert C == Cpsf, "PSF map should have the same channel as image"
    pad = int((ks - 1) / 2)
    img_pad = torch.nn.functional.pad(img, (pad, pad, pad, pad), mode="reflect")
    out = torch.zeros_like(img)
    for i in range(grid):
        for j in range(grid):
            k = torch.flip(psf_map[:, i * ks:(i + 1) * ks, j * ks:(j + 1) * ks], [1, 2]).unsqueeze(1)
            h0, w0 = int(i / grid * H), int(j / grid * W)
            h1, w1 = int((i + 1) / grid * H), int((j + 1) / grid * W)
            patch = img_pad[:, :, h0:h1 + 2 * pad, w0:w1 + 2 * pad]
            out[:, :, h0:h1, w0:w1] = torch.nn.functional.conv2d(patch, k, groups=C, padding="valid")
    return out
