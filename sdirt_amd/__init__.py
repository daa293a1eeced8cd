"""sdirt_amd -- MI355X-native dual-pixel ray-traced PSF renderer.

Drop-in for the PSF hot path of LinYark/Sdirt (deeplens/optics.py,
surfaces.py, monte_carlo.py): same Python call signatures, the per-ray work in
hand-written HIP kernels for gfx950 behind the C ABI of include/sdirt_dp.h.
"""
from . import _lib
from ._lib import SdirtError
from .basics import DEFAULT_WAVE, DEPTH, EPSILON, GEO_SPP, WAVE_RGB, Material, Ray
from .monte_carlo import (assign_points_to_pixels_big_r, assign_points_to_pixels_small_r,
                          forward_integral, forward_integral_lr)
from .optics import Lensgroup, PendingPSF
from .psfnet import PSFNet
from .dfdp import DfDPNet, dp_cost_volume
from .render_psf import (local_dp_psf_render, local_psf_render, local_psf_render_fast,
                         local_psf_render_high_res, render_psf, render_psf_map)
from .surfaces import Aspheric

__all__ = ["Lensgroup", "PSFNet", "DfDPNet", "dp_cost_volume", "PendingPSF", "Ray", "Material", "Aspheric", "SdirtError", "forward_integral",
           "forward_integral_lr", "assign_points_to_pixels_small_r",
           "assign_points_to_pixels_big_r", "local_psf_render", "local_psf_render_fast",
           "local_dp_psf_render", "local_psf_render_high_res", "render_psf", "render_psf_map", "DEFAULT_WAVE", "WAVE_RGB", "GEO_SPP", "EPSILON", "DEPTH"]
