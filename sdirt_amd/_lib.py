"""ctypes binding of libsdirt_dp.so (include/sdirt_dp.h).

The library is the product: there is NO CPU fallback.  If the shared object is
missing, `lib()` raises SdirtError telling how to build it; if no MI355X is
visible, the first device call fails with the HIP error text.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SDIRT_AMD_LIB overrides the path (kernel-variant A/B runs in tools/kbench.py only)
LIB_PATH = os.environ.get("SDIRT_AMD_LIB") or os.path.join(HERE, "libsdirt_dp.so")

ABI_VERSION = 4          # SDIRT_ABI_VERSION of include/sdirt_dp.h this binding was written against
MAX_SURFACES = 64
MAX_AI = 8
NEWTON_MAXITER = 10
MAX_KS = 141
MAX_KS_STAGED = 1024
MAX_WAVELENGTHS = 3
PSF_NORMALIZE = 1
PSF_STRICT_IEEE = 4
TRACE_NO_PREFETCH = 8
PSF_ONE_ROUND = 16
PSF_INTERLEAVED = 32
PSF_ZERO_CTL = 64
PSF_NO_VERIFY = 128
PSF_DETERMINISTIC = 256
CTL_STATUS, CTL_ANY_VALID, CTL_TRIPS2, CTL_MASKS, CTL_WORDS = 0, 1, 16, 64, 320
CTL_LANES = 1411
CTL_TAG = 2

KIND_PLANE, KIND_SPHERE, KIND_ASPHERE = 0, 1, 2


class SdirtError(RuntimeError):
    pass


class SurfaceDesc(C.Structure):
    """sdirt_surface_desc"""
    _fields_ = [("kind", C.c_int32), ("ai_degree", C.c_int32), ("r", C.c_double),
                ("d", C.c_float), ("c", C.c_float), ("k", C.c_float),
                ("ai", C.c_float * MAX_AI), ("n1", C.c_double), ("n2", C.c_double)]


class Rays(C.Structure):
    """sdirt_rays: eight device pointers"""
    _fields_ = [(n, C.c_void_p) for n in ("ox", "oy", "oz", "dx", "dy", "dz", "ra", "obliq")]


class DpParams(C.Structure):
    """sdirt_dp_params"""
    _fields_ = [("h", C.c_double), ("f", C.c_double), ("w", C.c_double), ("r", C.c_double)]


_P, _I64, _I32, _U32, _D = C.c_void_p, C.c_int64, C.c_int32, C.c_uint32, C.c_double

# name -> (restype, argtypes); mirrors include/sdirt_dp.h one to one
SIGNATURES = {
    "sdirt_abi_version": (C.c_int, []),
    "sdirt_last_error": (C.c_char_p, []),
    "sdirt_device_count": (C.c_int, []),
    "sdirt_lens_create": (C.c_int, [C.POINTER(SurfaceDesc), _I32, C.POINTER(_P)]),
    "sdirt_lens_destroy": (None, [_P]),
    "sdirt_lens_num_surfaces": (_I32, [_P]),
    "sdirt_points_to_object": (C.c_int, [_P, _I64, _D, _D, _D, _D, _P, _P]),
    "sdirt_pupil_samples": (C.c_int, [_P, _P, _I64, _D, _P, _P, _P]),
    "sdirt_sample_rays": (C.c_int, [_P, _I64, _P, _P, _I64, _D, Rays, _P]),
    "sdirt_rays_from_aos": (C.c_int, [_P, _P, _P, _I64, _I64, _I32, Rays, _P]),
    "sdirt_rays_to_aos": (C.c_int, [Rays, _I64, _I64, _P, _P, _P]),
    "sdirt_trace": (C.c_int, [_P, _I32, _I32, _I32, C.POINTER(_I32), _U32, Rays, _I64, _P, _P]),
    "sdirt_trace_to": (C.c_int, [_P, _I32, _I32, _I32, C.POINTER(_I32), _U32, Rays, Rays, _I64, _P, _P]),
    "sdirt_trace2sensor": (C.c_int, [_P, C.POINTER(_I32), _U32, _D, Rays, Rays, _I64, _P, _P]),
    "sdirt_propagate_to": (C.c_int, [_D, Rays, _I64, _P]),
    "sdirt_center_from_rays": (C.c_int, [Rays, _I64, _I64, _P, _P, _P]),
    "sdirt_forward_integral": (C.c_int, [Rays, _I64, _I64, _D, _I32, _P, C.POINTER(DpParams), _U32,
                                         _P, _P, _P]),
    "sdirt_forward_integral_plan": (C.c_int, [_I64, _I64, _I32, _I32, _I32, C.POINTER(_I64)]),
    "sdirt_psf_normalize": (C.c_int, [_P, _I64, _I32, _P]),
    "sdirt_chief_center": (C.c_int, [_P, _P, _I64, _P, _P, _I64, _D, _D, C.POINTER(_I32), _U32,
                                     _P, _P, _P, _P]),
    "sdirt_psf_lr": (C.c_int, [_P, _P, _I64, _P, _P, _I64, _D, _D, _D, _I32, _P,
                               C.POINTER(DpParams), C.POINTER(_I32), _U32, _P, _P, _P, _P]),
    "sdirt_psf_lr_centered": (C.c_int, [_P, _P, _P, _I64, _P, _P, _I64, _P, _P, _I64, _D, _D, _D, _I32,
                                        C.POINTER(DpParams), C.POINTER(_I32), C.POINTER(_I32), _U32,
                                        _P, _P, _P, _P, _P, _P, _P]),
    "sdirt_psf_rgb_centered": (C.c_int, [C.POINTER(_P), _I32, _P, _P, _I64, _P, _P, _I64, _P, _P, _I64, _D, _D, _D,
                                         _I32, C.POINTER(DpParams), C.POINTER(_I32), C.POINTER(_I32), _U32,
                                         _P, _P, _P, _P, _P, _P, _P]),
    "sdirt_psf_rgb": (C.c_int, [C.POINTER(_P), _I32, _P, _I64, _P, _P, _I64, _D, _D, _D, _I32, _P,
                                C.POINTER(DpParams), C.POINTER(_I32), _U32, _P, _P, _P, _P]),
    "sdirt_psf_spp_slices": (_I32, [_I64, _I64, _I32]),
    "sdirt_psf_verified_scratch_bytes": (_I64, [_I64, _I64]),
    "sdirt_psf_lr_verified": (C.c_int, [_P, _P, _P, _I64, _P, _P, _I64, _P, _P, _I64, _D, _D, _D, _I32,
                                        C.POINTER(DpParams), C.POINTER(_I32), C.POINTER(_I32), _U32,
                                        _P, _P, _P, _P, _P]),
    "sdirt_psf_call_scratch_bytes": (_I64, [_I64, _I64, _I64]),
    "sdirt_psf_call": (C.c_int, [_P, _P, _P, _I64, _P, _I64, _I64, _D, _D, _D, _D, _D, _I32, C.POINTER(DpParams),
                                 C.POINTER(_I32), C.POINTER(_I32), _U32, _P, _P, _P, _P, _P, _P]),
    "sdirt_ctl_to_lanes": (C.c_int, [_P, _U32, _P, _P]),
    "sdirt_ctl_from_lanes": (C.c_int, [_P, _P, C.POINTER(_I32), C.POINTER(_I32), _P, _P, _P]),
    "sdirt_host_uniform_fill": (C.c_int, [_P, _I64, _I64, _P]),
    "sdirt_selftest_math": (C.c_int, [_I32, C.c_uint64, C.c_uint64, _I32, _P, _P]),
    "sdirt_local_psf_render": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "sdirt_psfnet_render": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    "sdirt_mlp_packed_bytes": (_I64, [C.POINTER(_I32), _I32]),
    "sdirt_mlp_pack": (C.c_int, [C.POINTER(_P), C.POINTER(_P), C.POINTER(_I32), _I32, _P, _P]),
    "sdirt_psfnet_mlp": (C.c_int, [_P, C.POINTER(_I32), _I32, _P, _I64, _I32, _P, _P]),
    "sdirt_dp_cost_volume": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "sdirt_dp_cost_volume_nhwc": (C.c_int, [_P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "sdirt_tone_curve": (C.c_int, [_P, _I64, _I32, _P, _P]),
    "sdirt_avg_pool_windows": (C.c_int, [_P, _I64, _I32, _I32, _I32, _I32, _P, _P]),
    "sdirt_avg_pool_windows_nhwc": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "sdirt_bn_relu": (C.c_int, [_P, _I64, _I32, _I64, _P, _P, _P, _P, _I32, _I32, _P]),
    "sdirt_disparity_regression": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "sdirt_upsample_trilinear_ndhwc": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "sdirt_dp_cost_volume_backward": (C.c_int, [_P, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
}

class StreamArg:
    """A hipStream_t together with the GPU it belongs to.  ctypes passes `_as_parameter_`;
    the wrappers installed by lib() read `.index` to make that GPU the current HIP device for
    the duration of the call (a kernel launch goes to the CURRENT device: torch's default stream
    handle is 0 on every GPU, so the handle alone does not say where to run)."""
    __slots__ = ("_as_parameter_", "index")

    def __init__(self, handle, index):
        self._as_parameter_ = C.c_void_p(handle)
        self.index = index


class _Library:
    """Entry points of libsdirt_dp.so; every call whose stream argument is a StreamArg runs
    with that stream's GPU current (lens on cuda:1 while the caller's current device is cuda:0)."""

    def __init__(self, handle):
        self._handle = handle
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, self._guard(fn) if args and args[-1] is _P else fn)

    @staticmethod
    def _guard(fn):
        def call(*args):
            st = args[-1]
            if st.__class__ is StreamArg and st.index != _torch_cuda.current_device():
                with _torch_cuda.device(st.index):
                    return fn(*args)
            return fn(*args)
        call.__name__ = fn.__name__
        return call


_lib = None


class _LazyTorchCuda:
    """torch.cuda, imported on first use (this module loads without torch for header checks)."""
    def __getattr__(self, name):
        import torch
        globals()["_torch_cuda"] = torch.cuda
        return getattr(torch.cuda, name)


_torch_cuda = _LazyTorchCuda()


def lib():
    """The loaded library, with argtypes set.  Raises SdirtError if not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) and not os.environ.get("SDIRT_AMD_LIB"):
            # not built yet: try once with the in-tree Makefile (hipcc cross-compiles without a
            # GPU); anything else is a hard error -- there is no CPU implementation to fall back to
            import subprocess
            try:
                subprocess.check_call(["make", "-C", os.path.join(HERE, "csrc")],
                                      stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            except (OSError, subprocess.CalledProcessError):
                pass
        if not os.path.exists(LIB_PATH):
            raise SdirtError(
                f"{LIB_PATH} is missing: build it with `make -C sdirt_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "sdirt_amd has no CPU fallback.")
        try:
            h = C.CDLL(LIB_PATH)
        except OSError as e:      # pragma: no cover
            raise SdirtError(f"cannot load {LIB_PATH}: {e}") from e
        wrapped = _Library(h)
        if wrapped.sdirt_abi_version() != ABI_VERSION:
            raise SdirtError("libsdirt_dp.so ABI version mismatch; rebuild")
        _lib = wrapped
    return _lib


def check(rc):
    if rc != 0:
        msg = lib().sdirt_last_error().decode("utf-8", "replace")
        raise SdirtError(f"libsdirt_dp error {rc}: {msg}")


def device_count():
    return lib().sdirt_device_count()
