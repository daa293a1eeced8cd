"""Constants, materials and the device ray bundle.

Mirrors the parts of deeplens/basics.py that the PSF path reads: the module
constants (basics.py:18-36), Material.ior (basics.py:299-380) and Ray
(basics.py:216-296).  Ray keeps its state SoA on the GPU (one [7, M] fp32
buffer: ox oy oz dx dy dz ra; the obliquity factor `obliq`, which nothing on
the PSF path reads -- monte_carlo.py:46-50 computes and drops it --, in an
array of its own that exists only once somebody asks for it) and hands out the
reference's [..., 3] tensors only on request.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib

DEFAULT_WAVE = 0.589                     # basics.py:20
WAVE_RGB = [0.656, 0.589, 0.486]         # basics.py:22
DEPTH = -20000                           # basics.py:28
GEO_SPP = 2048                           # basics.py:29
EPSILON = 1e-9                           # basics.py:35
MAXT = 1e5                               # basics.py:33

#: True (default) = every Ray a CALLER makes -- Ray(...), Ray.from_normalized, Lensgroup.sample_from_points and the other
#: public samplers -- carries `obliq` from its construction on, as the reference's does (basics.py:240), and `ray.obliq`
#: can be read after any trace.  The bundles the package makes for ITSELF (the staged PSF chain for grids above 141
#: pixels: Ray.empty(obliq=False)) never carry it: nothing on the PSF path reads it, and a bundle without it moves 28
#: instead of 32 bytes per ray through every staged kernel.  False = callers' rays are lean as well: the array is
#: created by the first read or write of `ray.obliq`, which must then come BEFORE the trace.
TRACK_OBLIQ = True

_AIRLIKE = ("vacuum", "air", "occluder")


def default_device():
    return torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")


def require_gpu(device):
    if torch.device(device).type != "cuda":
        raise _lib.SdirtError(f"sdirt_amd runs on the GPU only (device='{device}' given); "
                              "there is no CPU fallback")


def stream_ptr(device=None):
    """torch's current stream on `device` as a ctypes argument that also names its GPU
    (_lib.StreamArg): the call it is passed to runs with that GPU as the current HIP device."""
    if device is not None:
        require_gpu(device)
    st = torch.cuda.current_stream(device)
    return _lib.StreamArg(st.cuda_stream, st.device.index)


def dptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


# user-registered dispersion formulas: name -> ("sellmeier"|"schott", 6 coefficients)
_REGISTERED = {}


def register_material(name, kind, coeffs, n=None, V=None):
    """Add a glass by Sellmeier (k1,l1,k2,l2,k3,l3) or Schott (a0..a5)
    coefficients (basics.py:105-146 holds the reference's catalogue; the lenses
    shipped with the reference only use 'air' and 'n/V' strings)."""
    if kind not in ("sellmeier", "schott") or len(coeffs) != 6:
        raise ValueError("kind must be 'sellmeier' or 'schott' with 6 coefficients")
    _REGISTERED[name.lower()] = (kind, [float(c) for c in coeffs], n, V)


class Material:
    """Refractive index model, basics.py:299-380.

    'air' / 'vacuum' / 'occluder'  -> Sellmeier with zero coefficients (n = 1)
    'n/V' string                   -> Cauchy: n = A + B / lambda_nm^2  (basics.py:336-338,355-362)
    registered name                -> Sellmeier or Schott formula
    """

    def __init__(self, name=None):
        self.name = "vacuum" if name is None else str(name).lower()
        if self.name in _AIRLIKE:
            self.dispersion, self.coeffs = "sellmeier", [0.0] * 6
            self.n, self.V = 1.0, math.inf
        elif self.name in _REGISTERED:
            self.dispersion, self.coeffs, self.n, self.V = _REGISTERED[self.name]
            if self.n is None:
                self.n, self.V = float(self.ior(0.5893)), math.inf
        else:
            try:
                n, V = self.name.split("/")
                self.n, self.V = float(n), float(V)
            except ValueError:
                raise ValueError(f"unknown material '{name}': use 'air', an 'n/V' pair or "
                                 "register_material()") from None
            self.dispersion, self.coeffs = "naive", None
        self.A, self.B = self.nV_to_AB(self.n, self.V)

    @staticmethod
    def nV_to_AB(n, V):
        # basics.py:355-362
        def ivs(a):
            return 1.0 / a ** 2
        lambdas = [656.3, 589.3, 486.1]
        B = (n - 1) / V / (ivs(lambdas[2]) - ivs(lambdas[0]))
        A = n - B * ivs(lambdas[1])
        return A, B

    def ior(self, wvln):
        """float64 refractive index at wvln [um] (values >= 10 are nm), basics.py:316-340."""
        wv = wvln if wvln < 10 else wvln * 1e-3
        if self.dispersion == "sellmeier":
            k1, l1, k2, l2, k3, l3 = self.coeffs
            n2 = 1 + k1 * wv ** 2 / (wv ** 2 - l1) + k2 * wv ** 2 / (wv ** 2 - l2) \
                + k3 * wv ** 2 / (wv ** 2 - l3)
            return float(np.sqrt(n2))
        if self.dispersion == "schott":
            a0, a1, a2, a3, a4, a5 = self.coeffs
            ws = wv ** 2
            n2 = a0 + a1 * ws + (a2 + (a3 + (a4 + a5 / ws) / ws) / ws) / ws
            return float(np.sqrt(n2))
        return self.A + self.B / (wv * 1e3) ** 2


class Ray:
    """A bundle of rays with one wavelength, resident on the GPU (basics.py:216-296).

    Ray(o, d, wvln, ra=...) accepts the reference's [..., 3] tensors, moves them
    to `device` and L2-normalises d (basics.py:244-245) in a HIP kernel.
    """

    def __init__(self, o, d, wvln=DEFAULT_WAVE, normalized=True, ra=None, en=None, obliq=None,
                 opl=None, coherent=False, device=None):
        if coherent:
            raise NotImplementedError("coherent ray tracing is outside the DP-PSF path")
        device = torch.device(device) if device is not None else default_device()
        o = o if torch.is_tensor(o) else torch.tensor(o)
        d = d if torch.is_tensor(d) else torch.tensor(d)
        o = o.to(device=device, dtype=torch.float32).contiguous()
        d = d.to(device=device, dtype=torch.float32).expand_as(o).contiguous()
        self._init_empty(tuple(o.shape[:-1]), wvln, device)
        ra_d = None
        if ra is not None:
            ra_d = ra.to(device=device, dtype=torch.float32).expand(self.shape).contiguous()
        _lib.check(_lib.lib().sdirt_rays_from_aos(dptr(o), dptr(d), dptr(ra_d), self.numel, self._n_points(), 1,
                                                  self.c_rays(), stream_ptr(device)))
        if obliq is not None:
            self.obliq = obliq

    @classmethod
    def from_normalized(cls, o, d, wvln=DEFAULT_WAVE, ra=None, device=None):
        """Adopt unit direction vectors exactly as given (the constructor re-normalises, as the
        reference's does; renormalising an already normalised vector can move its last bit).
        For ray-level hand-off: rays captured AFTER the reference's Ray.__init__ (basics.py:245)."""
        device = torch.device(device) if device is not None else default_device()
        o = torch.as_tensor(o).to(device=device, dtype=torch.float32).contiguous()
        d = torch.as_tensor(d).to(device=device, dtype=torch.float32).expand_as(o).contiguous()
        self = cls.__new__(cls)
        self._init_empty(tuple(o.shape[:-1]), wvln, device)
        ra_d = None
        if ra is not None:
            ra_d = ra.to(device=device, dtype=torch.float32).expand(self.shape).contiguous()
        _lib.check(_lib.lib().sdirt_rays_from_aos(dptr(o), dptr(d), dptr(ra_d), self.numel, self._n_points(), 0,
                                                  self.c_rays(), stream_ptr(device)))
        return self

    # -- construction helpers -------------------------------------------------
    def _init_empty(self, shape, wvln, device, track=None):
        require_gpu(device)
        self.shape = tuple(int(s) for s in shape)
        self.numel = int(np.prod(self.shape)) if len(self.shape) else 1
        self.wvln = wvln if wvln < 10 else wvln * 1e-3      # basics.py:235
        self.coherent = False
        self.device = torch.device(device)
        self.soa = torch.empty((7, max(self.numel, 1)), dtype=torch.float32, device=self.device)
        # the obliquity factor: its own [M] array, or None = "all ones so far, not stored"; _ob_lost: the bundle has
        # been traced without it (the products of surfaces.py:674 were not kept)
        self._ob = None
        self._ob_lost = False
        if TRACK_OBLIQ if track is None else track:
            self._ob = torch.ones(max(self.numel, 1), dtype=torch.float32, device=self.device)

    @classmethod
    def empty(cls, shape, wvln=DEFAULT_WAVE, device=None, obliq=False):
        """An uninitialised bundle WITHOUT an obliquity array (whatever TRACK_OBLIQ says: this is the constructor of the
        package's own bundles); obliq=True: with an (uninitialised) one."""
        self = cls.__new__(cls)
        self._init_empty(shape, wvln, device if device is not None else default_device(), track=False)
        if obliq:
            self._ob = torch.empty(max(self.numel, 1), dtype=torch.float32, device=self.device)
        return self

    @property
    def has_obliq(self):
        return self._ob is not None

    def _adopt(self, traced):
        """Take over the storage of `traced` (the out-of-place result of a trace of THIS bundle): the reference's
        trace rebinds ray.o / ray.d / ray.ra to new tensors in the same way (surfaces.py:425, 676-677).  Views
        (`ray.ra`, `ray.obliq`) and `c_rays()` pointers taken BEFORE the trace keep showing the untraced bundle."""
        self.soa, self._ob = traced.soa, traced._ob
        self._mark_traced()

    def _mark_traced(self):
        if self._ob is None:
            self._ob_lost = True

    def _n_points(self):
        """Order of the rays in `soa` (include/sdirt_dp.h, sdirt_rays): a 2-D bundle [spp, N] -- the shape of every
        bundle on the PSF path -- is stored POINT-MAJOR, ray (s, n) at n * spp + s, so that the per-point
        reductions (centroid, splat) read each point's rays as one contiguous run; bundles of any other rank are
        stored in the row-major order of their shape.  -> N for the former, 1 for the latter."""
        return self.shape[1] if len(self.shape) == 2 else 1

    def _field(self, row):
        """Row `row` of the SoA buffer (7: the obliquity array) as a tensor of the reference's shape (a view: writes
        go through)."""
        flat = (self.soa[row] if row < 7 else self._obliq_storage())[:self.numel]
        if len(self.shape) == 2:
            return flat.view(self.shape[1], self.shape[0]).t()
        return flat.view(self.shape)

    def _set_field(self, row, v):
        self._field(row).copy_(v.to(self.device, torch.float32).expand(self.shape))

    def c_rays(self):
        """The bundle as the C ABI takes it (include/sdirt_dp.h: sdirt_rays); obliq = NULL unless the array exists."""
        base, stride = self.soa.data_ptr(), self.soa.stride(0) * 4
        return _lib.Rays(*([C.c_void_p(base + i * stride) for i in range(7)] + [dptr(self._ob)]))

    def _obliq_storage(self):
        """The obliquity array, created on first use: all ones for a bundle that has not been traced yet (what the
        reference's constructor sets, basics.py:240); a bundle that HAS been traced without it cannot make it up."""
        if self._ob is None:
            if self._ob_lost:
                raise _lib.SdirtError(
                    "ray.obliq: this bundle was traced without its obliquity factor (nothing on the PSF path reads it, so "
                    "it is only carried on request): read or set ray.obliq BEFORE tracing, construct the Ray with "
                    "obliq=..., or set sdirt_amd.basics.TRACK_OBLIQ = True to carry it always, as the reference does")
            self._ob = torch.ones(max(self.numel, 1), dtype=torch.float32, device=self.device)
        return self._ob

    # -- reference-shaped views -----------------------------------------------
    def _aos(self, which):
        out = torch.empty(self.shape + (3,), dtype=torch.float32, device=self.device)
        args = (dptr(out), None) if which == "o" else (None, dptr(out))
        _lib.check(_lib.lib().sdirt_rays_to_aos(self.c_rays(), self.numel, self._n_points(), *args,
                                                stream_ptr(self.device)))
        return out

    def _set_aos(self, row0, value):
        v = value.to(device=self.device, dtype=torch.float32).expand(self.shape + (3,))
        for c in range(3):
            self._field(row0 + c).copy_(v[..., c])

    @property
    def o(self):
        return self._aos("o")

    @o.setter
    def o(self, v):
        self._set_aos(0, v)

    @property
    def d(self):
        return self._aos("d")

    @d.setter
    def d(self, v):
        self._set_aos(3, v)

    @property
    def ra(self):
        return self._field(6)

    @ra.setter
    def ra(self, v):
        self._set_field(6, v)

    @property
    def obliq(self):
        return self._field(7)

    @obliq.setter
    def obliq(self, v):
        if self._ob is None:
            self._ob_lost = False
            self._ob = torch.empty(max(self.numel, 1), dtype=torch.float32, device=self.device)
        self._set_field(7, v)

    @property
    def en(self):          # never modified on this path (basics.py:239)
        return torch.ones(self.shape, dtype=torch.float32, device=self.device)

    @property
    def opl(self):         # only used by coherent tracing (basics.py:241)
        return torch.zeros(self.shape, dtype=torch.float32, device=self.device)

    phi = opl

    # -- methods ----------------------------------------------------------------
    def propagate_to(self, z, n=1):
        """basics.py:256-274 (incoherent part), in place."""
        _lib.check(_lib.lib().sdirt_propagate_to(float(z), self.c_rays(), self.numel,
                                                 stream_ptr(self.device)))
        return self

    prop_to = propagate_to

    def project_to(self, z):
        """basics.py:277-285: intersection with plane z, [..., 2]; does not move the rays."""
        c = self.clone()
        c.propagate_to(z)
        return c.o[..., :2]

    def clone(self, device=None):
        c = Ray.empty(self.shape, self.wvln, self.device if device is None else device)
        c.soa.copy_(self.soa)
        c._ob = None if self._ob is None else self._ob.to(c.device, copy=True)
        c._ob_lost = self._ob_lost
        return c

    def to(self, device):
        device = torch.device(device)
        if device != self.device:
            self.soa = self.soa.to(device)
            self._ob = None if self._ob is None else self._ob.to(device)
            self.device = device
        return self
