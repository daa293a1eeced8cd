"""Surface records of a lens prescription.

The reference models every surface of its JSON lenses as `Aspheric`
(deeplens/surfaces.py:281-331) and picks the plane / sphere / asphere code path
at trace time (surfaces.py:409,456,491).  Here a surface is a plain record; the
branch is decided once (`kind`) and the intersection / refraction math runs in
the HIP kernels (sdirt_amd/csrc/sdirt_device.hpp).
"""
import numpy as np

from . import _lib
from .basics import Material

KIND_NAMES = {_lib.KIND_PLANE: "plane", _lib.KIND_SPHERE: "sphere", _lib.KIND_ASPHERE: "asphere"}


class Aspheric:
    """r: semi-aperture (python float), d: vertex z, c: curvature, k: conic,
    ai: even-asphere coefficients [ai2, ai4, ...] or None (all fp32, as the
    reference stores them in fp32 tensors)."""

    def __init__(self, r, d, c=0.0, k=0.0, ai=None, mat1=None, mat2=None, device=None):
        self.r = float(r)
        self.d = np.float32(d)
        self.c = np.float32(c)
        self.k = np.float32(k)
        if ai is not None:
            if len(ai) > _lib.MAX_AI:
                raise ValueError(f"at most {_lib.MAX_AI} aspheric terms are supported")
            self.ai = np.asarray(ai, dtype=np.float32)
            self.ai_degree = len(ai)
        else:
            self.ai = None
            self.ai_degree = 0
        self.mat1 = mat1 if isinstance(mat1, Material) else Material(mat1)
        self.mat2 = mat2 if isinstance(mat2, Material) else Material(mat2)

    @property
    def kind(self):
        if float(self.c) == 0.0:                                  # surfaces.py:409
            return _lib.KIND_PLANE
        if self.ai is None and float(self.k) == 0.0:              # surfaces.py:456
            return _lib.KIND_SPHERE
        return _lib.KIND_ASPHERE

    def desc(self, wvln):
        """sdirt_surface_desc at one wavelength."""
        s = _lib.SurfaceDesc()
        s.kind = self.kind
        s.ai_degree = self.ai_degree if self.kind == _lib.KIND_ASPHERE else 0
        s.r = self.r
        s.d, s.c, s.k = float(self.d), float(self.c), float(self.k)
        for i in range(s.ai_degree):
            s.ai[i] = float(self.ai[i])
        s.n1 = float(self.mat1.ior(wvln))
        s.n2 = float(self.mat2.ior(wvln))
        return s

    def ray_reaction(self, ray):
        """surfaces.py:391-520: intersect `ray` with THIS surface and refract it; the ray object is updated and also
        returned, as in the reference (its storage is REBOUND to the traced bundle, like the reference's ray.o / d / ra:
        views and c_rays() pointers taken before the call keep showing the untraced rays -- re-fetch them).  Newton intersection with the batch-wide trip count of the reference's loop
        (surfaces.py:547, speculated and verified like Lensgroup.trace does it), position / weight update of the rays
        that hit inside the aperture, refraction unless the surface is a plane between equal media.  The direction
        of travel is the reference's test, sum(d_z * ra) > 0 (surfaces.py:399-405: n1/n2 forward, n2/n1 backward).
        One launch of sdirt_trace_to on a one-surface table at the ray's wavelength (cached on the surface)."""
        import ctypes as C

        import torch

        from .basics import Ray, dptr, stream_ptr
        from .newton import TripPlanner
        from .optics import _DevLens
        dev = ray.device
        key = (float(ray.wvln), dev.index)
        tables = self.__dict__.setdefault("_dev_tables", {})
        state = (self.kind, self.r, float(self.d), float(self.c), float(self.k), None if self.ai is None else tuple(self.ai.tolist()),
                 self.mat1.name, self.mat2.name)
        if tables.get(key, (None, None))[1] != state:       # first use, or the record was edited since
            with torch.cuda.device(dev):
                tables[key] = (_DevLens([self], float(ray.wvln)), state)
            self.__dict__["_planner"] = TripPlanner()
        handle = tables[key][0].handle
        planner = self.__dict__.setdefault("_planner", TripPlanner())
        forward = bool(float((ray._field(5) * ray.ra).sum()) > 0)
        curved = [self.kind != _lib.KIND_PLANE]
        mask = torch.zeros(_lib.MAX_SURFACES, dtype=torch.int32, device=dev)
        dst = Ray.empty(ray.shape, ray.wvln, dev, obliq=ray.has_obliq)   # out of place: a re-launch with a corrected count re-reads `ray`

        def launch(trips):
            mask.zero_()
            _lib.check(_lib.lib().sdirt_trace_to(handle, 0, 1, 0 if forward else 1, (C.c_int32 * 1)(int(trips[0])), 0,
                                                 ray.c_rays(), dst.c_rays(), ray.numel, dptr(mask), stream_ptr(dev)))
            return mask[:1].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        planner.run(("ray_reaction", forward), curved, [0], launch)
        ray._adopt(dst)                                     # the reference rebinds ray.o / ray.d / ray.ra as well (:425, :676)
        return ray

    def surface(self, x, y):
        """surfaces.py:766-771 with :787-808, on the host in fp32 numpy: the sag z(x, y) of the surface about its
        vertex, evaluated at the apex where (x, y) lies outside the conic's domain (_valid_loose, :735-743).  For
        drawing and for setting up rays; the tracing kernels carry their own evaluation (sdirt_device.hpp)."""
        x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)
        rr = x * x + y * y
        if float(self.c) == 0.0:
            inside = np.ones_like(rr, dtype=bool)
        elif float(self.k) > -1:
            with np.errstate(divide="ignore"):
                inside = rr < np.float32((1 - 1e-9) / float(self.c) ** 2 / (1 + float(self.k)))
        else:
            inside = rr > 0
        r2 = np.where(inside, rr, np.float32(0))
        z = r2 * self.c / (1 + np.sqrt(1 - (1 + self.k) * r2 * self.c ** 2, dtype=np.float32))
        if self.ai is not None:
            for i, a in enumerate(self.ai):
                z = z + a * r2 ** (i + 1)
        return z.astype(np.float32)

    def surface_with_offset(self, x, y):
        """surfaces.py:172-175: the surface's z in lens coordinates."""
        return self.surface(x, y) + self.d

    def surf_dict(self):
        """Same keys as the reference's JSON surfaces (optics.py:2155-2167)."""
        kind = self.kind
        d = {"type": {0: "Stop", 1: "Spheric", 2: "Aspheric"}[kind], "r": self.r,
             "c": float(self.c), "d": float(self.d), "mat1": self.mat1.name,
             "mat2": self.mat2.name}
        if kind == _lib.KIND_ASPHERE:
            d["k"] = float(self.k)
            d["ai"] = [float(a) for a in (self.ai if self.ai is not None else [])]
        return d

    def __repr__(self):
        return (f"Aspheric({KIND_NAMES[self.kind]}, r={self.r}, d={float(self.d)}, "
                f"c={float(self.c)}, k={float(self.k)}, ai={self.ai}, "
                f"{self.mat1.name}->{self.mat2.name})")
