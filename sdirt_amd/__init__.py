"""sdirt_amd -- MI355X-native dual-pixel ray-traced PSF renderer.

Drop-in for the PSF hot path of LinYark/Sdirt (deeplens/optics.py,
surfaces.py, monte_carlo.py): same Python call signatures, the per-ray work in
hand-written HIP kernels for gfx950 behind the C ABI of include/sdirt_dp.h.
"""
import os as _os

# Two process-environment settings a host program built on this package wants (they lived in bench.py only).
# Both are read when the runtime in question initialises, so they are set here, at import, and only where the caller has
# not decided otherwise (SDIRT_NO_ENV_DEFAULTS=1: hands off):
#   GPU_MAX_HW_QUEUES=16  HIP maps a process's streams onto 4 hardware queues by default and a hardware queue runs in order;
#                         the render loop of a sharded volume uses five streams and three communicators, and on four queues
#                         a step's kernel waits behind the previous step's mask all-reduce (profiles/r05/sweep_hw_queues.txt)
#   OMP_WAIT_POLICY=passive  idle OpenMP workers of torch's CPU ops otherwise spin a cgroup's CPU quota away and the calling
#                         thread is throttled for tens of milliseconds in its next blocking wait (profiles/r05/tcp_span_omp_spinning.txt)
if _os.environ.get("SDIRT_NO_ENV_DEFAULTS") != "1":
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    _os.environ.setdefault("OMP_WAIT_POLICY", "passive")

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
