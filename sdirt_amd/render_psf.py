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


def local_psf_render_high_res(input, psf, patch_size=[320, 480], kernel_size=11):
    """render_psf.py:191-208: the image cut into patch_size tiles, every tile rendered on its own
    by local_psf_render -- so each tile is replicate-padded at ITS OWN border.  Returns (rl, rr)
    [N,C,H,W].  (The reference's version assigns local_psf_render's (left, right) tuple into one
    tensor and raises TypeError; this is what its loop computes with both halves kept, pinned by
    fixture F18 = the reference's local_psf_render applied tile by tile.)"""
    _, _, height, width = input.shape
    rl, rr = torch.zeros_like(input), torch.zeros_like(input)
    for top in range(0, height, patch_size[0]):
        for left in range(0, width, patch_size[1]):
            rows = slice(top, min(top + patch_size[0], height))
            cols = slice(left, min(left + patch_size[1], width))
            rl[:, :, rows, cols], rr[:, :, rows, cols] = local_psf_render(
                input[:, :, rows, cols], psf[:, rows, cols], kernel_size=kernel_size)
    return rl, rr


def _depthwise_convolution(padded, kernels):
    """True (flipped-kernel) 2-D convolution of every channel with its own [ks, ks] kernel, no
    further padding: stock depthwise conv2d (MIOpen), which correlates, on the flipped kernels."""
    flipped = kernels.flip(-2, -1).unsqueeze(1)                     # [C, 1, ks, ks]
    return torch.nn.functional.conv2d(padded, flipped, groups=padded.shape[1])


def render_psf(img, psf):
    """render_psf.py:12-28: one PSF per channel for the whole image, [B,C,H,W] * [C,ks,ks], reflect
    padding.  Dense depthwise convolution: library work, no custom kernel."""
    half = psf.shape[-1] // 2
    return _depthwise_convolution(torch.nn.functional.pad(img, (half,) * 4, mode="reflect"), psf)


def render_psf_map(img, psf_map, grid):
    """render_psf.py:31-73: psf_map [C, grid*ks, grid*ks] holds one PSF per image tile (tile (i, j)
    covers rows int(i/grid*H) .. int((i+1)/grid*H) and the matching columns); every tile is
    convolved with its PSF, reading across the tile border into the (reflect-padded) image."""
    assert img.dim() == 4, "Input image should be [B, C, H, W]"
    channels, map_h, map_w = psf_map.shape
    assert map_h % grid == 0 and map_w % grid == 0, "PSF map size should be divisible by grid"
    ks = map_h // grid
    assert ks % 2 == 1, "PSF kernel size should be odd"
    _, c, height, width = img.shape
    assert c == channels, "PSF map should have the same channel as image"
    half = (ks - 1) // 2
    padded = torch.nn.functional.pad(img, (half,) * 4, mode="reflect")
    tiles = psf_map.view(channels, grid, ks, grid, ks).permute(1, 3, 0, 2, 4)    # [grid, grid, C, ks, ks]
    row_edge = [int(i / grid * height) for i in range(grid + 1)]
    col_edge = [int(j / grid * width) for j in range(grid + 1)]
    out = torch.zeros_like(img)
    for i in range(grid):
        for j in range(grid):
            window = padded[:, :, row_edge[i]:row_edge[i + 1] + 2 * half, col_edge[j]:col_edge[j + 1] + 2 * half]
            out[:, :, row_edge[i]:row_edge[i + 1], col_edge[j]:col_edge[j + 1]] = \
                _depthwise_convolution(window, tiles[i, j])
    return out
