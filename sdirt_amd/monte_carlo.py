"""Forward Monte-Carlo integral: sensor-plane rays -> dual-pixel PSF grids.

Same call signatures as deeplens/monte_carlo.py (forward_integral :9,
assign_points_to_pixels_small_r :135, _big_r :242); the work is one HIP kernel
(sdirt_forward_integral): window test, closed-form left/right sub-pixel areas
and the 4-tap bilinear scatter-add.
"""
import ctypes as C

import torch

from . import _lib
from .basics import Ray, dptr, stream_ptr


def _dp(param_list):
    if param_list is None:
        return None, "l"
    h, f, w, r, direct = param_list
    return _lib.DpParams(float(h), float(f), float(w), float(r)), direct


def _flags(precision):
    if precision not in ("lean", "ieee"):
        raise ValueError("precision must be 'lean' or 'ieee'")
    return _lib.PSF_STRICT_IEEE if precision == "ieee" else 0


def forward_integral_lr(ray, ps, ks, pointc_ref=None, param_list=None, precision="lean"):
    """RAW (l_grid, r_grid), each [N, ks, ks]; r_grid is all-zero when
    param_list is None exactly as in monte_carlo.py:230-235.  precision='ieee': the reference's literal
    sequence for the sub-pixel areas (arccos, sin) and the compiler's full-range divisions instead of the
    fused segment-area polynomial (SDIRT_PSF_STRICT_IEEE)."""
    if len(ray.shape) != 2:
        raise ValueError("ray must have shape [spp, N]")
    S, N = ray.shape
    dev = ray.device
    if pointc_ref is None:
        # RMS centre, monte_carlo.py:28-31
        center = torch.empty((N, 2), dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().sdirt_center_from_rays(ray.c_rays(), S, N, dptr(center), None,
                                                     stream_ptr(dev)))
    else:
        center = pointc_ref.to(dev, torch.float32).reshape(N, 2).contiguous()
    dp, _ = _dp(param_list)
    lg = torch.empty((N, ks, ks), dtype=torch.float32, device=dev)
    rg = torch.empty_like(lg)
    _lib.check(_lib.lib().sdirt_forward_integral(
        ray.c_rays(), S, N, float(ps), int(ks), dptr(center),
        C.byref(dp) if dp is not None else None, _flags(precision), dptr(lg), dptr(rg), stream_ptr(dev)))
    return lg, rg


def forward_integral(ray, ps, ks, pointc_ref=None, interpolate=False, param_list=None, precision="lean"):
    """monte_carlo.py:9-68 -> [N, ks, ks]: the left grid, or the right one when
    param_list[4] != 'l' (the reference returns `psf_l` of a swapped pair, :64,237-240)."""
    lg, rg = forward_integral_lr(ray, ps, ks, pointc_ref, param_list, precision)
    _, direct = _dp(param_list)
    return lg if direct == "l" else rg


def _assign(points, ks, x_range, ra, x_tan, param_list, big, precision="lean"):
    if param_list is None:
        r = 0.5
    else:
        r = param_list[3]
    assert (r >= 0.5) if big else (r <= 0.5)
    dev = points.device
    if dev.type != "cuda":
        raise _lib.SdirtError("sdirt_amd splats on the GPU only (no CPU fallback)")
    S = points.shape[0]
    ps = (x_range[1] - x_range[0]) / (ks - 1)
    # forward_integral negates o and divides -d_x by d_z: feed o = -points,
    # d = (-x_tan, 0, 1) so that the kernel sees exactly `points` and `x_tan`.
    ray = Ray.empty((S, 1), device=dev)
    pts = points.to(torch.float32).reshape(S, 2)
    ray.soa[0, :S] = -pts[:, 0]
    ray.soa[1, :S] = -pts[:, 1]
    ray.soa[2, :S] = 0.0
    ray.soa[3, :S] = -x_tan.to(torch.float32).reshape(S)
    ray.soa[4, :S] = 0.0
    ray.soa[5, :S] = 1.0
    ray.soa[6, :S] = ra.to(torch.float32).reshape(S)
    center = torch.zeros((1, 2), dtype=torch.float32, device=dev)
    if big and param_list is None:
        param_list = [0.78, 1.44, 0.3, 0.5, "l"]
    lg, rg = forward_integral_lr(ray, ps, ks, center, param_list, precision)
    lg, rg = lg[0], rg[0]
    direct = "l" if param_list is None else param_list[4]
    return (lg, rg) if direct == "l" else (rg, lg)


def assign_points_to_pixels_small_r(points, ks, x_range, y_range, ra, interpolate=True,
                                    coherent=False, phase=None, d=None, obliq=None, wvln=0.589,
                                    x_tan=None, param_list=None, precision="lean"):
    """monte_carlo.py:135-240 for one point source: points [spp,2] inside the
    PSF window, ra [spp], x_tan [spp] -> (l_grid, r_grid) [ks,ks]."""
    return _assign(points, ks, x_range, ra, x_tan, param_list, big=False, precision=precision)


def assign_points_to_pixels_big_r(points, ks, x_range, y_range, ra, interpolate=True,
                                  coherent=False, phase=None, d=None, obliq=None, wvln=0.589,
                                  x_tan=None, param_list=None):
    """monte_carlo.py:242-372 (microlens radius r >= 0.5)."""
    return _assign(points, ks, x_range, ra, x_tan, param_list, big=True)
