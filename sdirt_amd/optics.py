"""Lensgroup: the reference's PSF-by-ray-tracing API on top of libsdirt_dp.so.

Same method names, argument meaning and defaults as deeplens/optics.py for the
dual-pixel PSF path (psf / psf_diff / psf_rgb / psf_map / psf_center /
sample_from_points / trace / trace2sensor / point_source_grid /
calc_scale_pinhole / entrance_pupil / exit_pupil / refocus / calc_fov /
read_lens_json).  All per-ray work runs in HIP kernels; this file only
validates arguments, draws the pupil uniforms from torch's CPU generator in the
reference's order (optics.py:483-484: that is what makes seeds reproduce),
keeps the speculated Newton trip tables (newton.py) and launches.

There is no CPU fallback: every method that traces rays raises if
libsdirt_dp.so is not built or no MI355X is visible.
"""
import ctypes as C
import json

import weakref

import numpy as np
import torch

from . import _hostrng, _lib, basics
from .basics import DEFAULT_WAVE, DEPTH, GEO_SPP, WAVE_RGB, Ray, dptr, stream_ptr
from .newton import NEWTON_MAXITER, TripPlanner
from .surfaces import Aspheric


def _as_device(device):
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
    return device


class _DevLens:
    """Owner of one sdirt_lens handle (a prescription at one wavelength)."""

    def __init__(self, surfaces, wvln):
        arr = (_lib.SurfaceDesc * len(surfaces))(*[s.desc(wvln) for s in surfaces])
        h = C.c_void_p()
        _lib.check(_lib.lib().sdirt_lens_create(arr, len(surfaces), C.byref(h)))
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().sdirt_lens_destroy(self.handle)
                self.handle = None
        except Exception:      # interpreter shutdown
            pass


class _EventBracket:
    """Records a (start, end) HIP event pair on torch's current stream -- the
    stream the kernels are launched on -- when profiling is switched on."""

    def __init__(self, sink, name, device):
        self.sink, self.name, self.device = sink, name, device

    def __enter__(self):
        if self.sink is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream(self.device))
        return self

    def __exit__(self, *exc):
        if self.sink is not None:
            self.e1.record(torch.cuda.current_stream(self.device))
            self.sink.setdefault(self.name, []).append((self.e0, self.e1))
        return False


class PendingPSF:
    """A psf_lr call whose kernel is enqueued and whose Newton trip check is still due."""

    def __init__(self, finish):
        self._finish, self._result = finish, None

    def wait(self):
        if self._finish is not None:
            self._result = self._finish()
            self._finish = None
        return self._result


class Lensgroup:
    """optics.py:22-116.  `device` must be a CUDA (ROCm) device for anything that traces."""

    #: psf_lr(defer=True) of the fitting shape (few points, many samples) through ONE library call (sdirt_psf_call, as the
    #: synchronous call) instead of the general path's separate launches.  Alone it is the faster producer -- 0.25 against
    #: 0.30 ms per 64 x 20000 batch with two in flight, 0.18 against 0.27 ms of host work (tools/psf_defer_ab.py).  Beside
    #: the fitting loop's hipGraph it is bimodal (tools/fit_ab.py, six fresh train_psfnet calls per setting): 0.73 ms per
    #: iteration against the general path's 0.79 in most calls, 1.03-1.09 in the others -- and in three runs out of three
    #: inside bench.py's process -- when the PSF batches and the step end up serialised (a batch's workgroups are one
    #: generation that fills every wave slot of the chip; the general path's separate launches leave the gaps the step's
    #: small kernels slip through).  Off by default: train_psfnet, the caller of defer=True, keeps the path whose worst case
    #: is the better one; a pure data generator may switch it on.
    defer_one_call = False

    def __init__(self, filename=None, sensor_res=(1024, 1024), use_roc=False,
                 post_computation=True, device=None):
        self.device = _as_device(device)
        self.surfaces = []
        self.materials = []
        self.sensor_res = sensor_res
        self.aper_idx = None
        self._dev = {}                 # wvln -> _DevLens
        self._pupil_cache = {}         # entrance(bool) -> (z, r)
        self.trips = TripPlanner()
        #: 'reference' = reproduce the reference's batch-global Newton trip counts
        #: (speculate + verify, newton.py); 'max' = always 10 trips, no host sync;
        #: 'adaptive' = speed mode: every 64-ray wave stops iterating when none of its rays is
        #: open (<= 10 trips), no host sync -- not batch-exact (PSFs differ by ~1e-5 of peak).
        self.trip_policy = "reference"
        #: optional hook reducing convergence masks over ranks (set by sdirt_amd.dist)
        self.mask_reduce = None
        #: 'lean' (default): division / sqrt by rcp/sqrt seed + fma corrections, proven
        #: bit-identical to IEEE for normal-range operands (all 2^46 mantissa pairs / every
        #: fp32 >= 2^-100; tools/selftest_math.py).  'ieee': the compiler's full-range
        #: sequences (also handles denormal / zero-denominator operands), ~1.26x slower.
        self.precision = "lean"
        #: how the paraxial pupil is estimated from the 16 traced rays (optics.py:1470-1515):
        #: 'reference' = the reference's estimator: pairwise 2x2 systems solved in fp32 by
        #: torch.linalg.lstsq on the host (the same LAPACK call the reference's CPU path
        #: makes).  It is ill-conditioned in fp32 and lands 0.11 % above the exact value on
        #: rf50mm -- the reference's PSFs are rendered with THAT pupil, so it is the default.
        #: 'exact' = closed-form float64 intersections (deterministic, unbiased).
        self.pupil_method = "reference"
        #: where the uniform -> pupil-disc mapping runs: 'device' (default) or 'host' (the
        #: reference's own torch CPU expressions; see _pupil_samples)
        self.pupil_mapping = "device"
        #: generator device of Lensgroup.sample_pupil (the per-source pupil samples of sample_point_source, analysis_rms,
        #: calc_magnification3): None = the lens's device, as the reference draws them; 'cpu' = the CPU generator
        #: (what a CPU run of the reference uses: seeds then reproduce its sample points)
        self.sample_rng_device = None
        #: True: psf calls ask for run-to-run IDENTICAL grids also where the default sums them in fp32 LDS tiles (grids of 50
        #: to 70 pixels -- config 2's 65 x 65; L alone: to 99): SDIRT_PSF_DETERMINISTIC, float64 tiles in 1024-thread
        #: workgroups, +0.5 % time.  Grids up to 49 pixels are summed in float64 anyway.  Beyond the range, with r > 0.5 or
        #: few points with many samples (the spp axis cut) the library says SDIRT_ERR_UNSUPPORTED.
        self.deterministic = False
        #: when a dict, kernel launches are bracketed with HIP events on the launch
        #: stream: {'psf_lr': [(start, end), ...], 'chief_center': [...]}  (bench.py)
        self.kernel_events = None
        if filename is not None:
            self.lens_name = filename
            self.load_file(filename, use_roc, sensor_res, post_computation)

    # ------------------------------------------------------------------ init
    def load_file(self, filename, use_roc=False, sensor_res=(1024, 1024), post_computation=True):
        """optics.py:118-142 (JSON only; the .txt reader is undefined in the reference)."""
        if filename.endswith(".json"):
            self.read_lens_json(filename)
        else:
            raise Exception("File format not supported.")
        self.find_aperture()
        self.prepare_sensor(sensor_res)
        self.diff_surf_range = self.find_diff_surf()
        if post_computation:
            self.post_computation()

    def read_lens_json(self, filename="./test.json"):
        """optics.py:2173-2198.  Also accepts this package's flat schema
        (sdirt_amd/data/*.json: keys kind/semi_aperture/z/curvature/...)."""
        with open(filename, "r") as f:
            data = json.load(f)
        self.surfaces, self.materials = [], []
        for sd in data["surfaces"]:
            if "type" in sd:
                t = sd["type"]
                if t == "Aspheric":
                    s = Aspheric(r=sd["r"], d=sd["d"], c=sd["c"], k=sd["k"], ai=sd["ai"],
                                 mat1=sd["mat1"], mat2=sd["mat2"])
                elif t in ("Stop", "Spheric"):
                    s = Aspheric(r=sd["r"], d=sd["d"], c=sd["c"], mat1=sd["mat1"], mat2=sd["mat2"])
                else:
                    raise Exception("Surface type not implemented.")
            else:
                ai = sd.get("even_asphere")
                s = Aspheric(r=sd["semi_aperture"], d=sd["z"], c=sd["curvature"],
                             k=sd.get("conic", 0.0), ai=ai if sd["kind"] == "asphere" else None,
                             mat1=sd["glass_before"], mat2=sd["glass_after"])
            self.surfaces.append(s)
            self.materials.append(s.mat1)
        self.materials.append(self.surfaces[-1].mat2)
        self.r_last = data["r_last"]
        self.d_sensor = data["d_sensor"]
        self._invalidate()

    def write_lens_json(self, filename="./test.json"):
        """optics.py:2145-2170."""
        data = {"foclen": getattr(self, "foclen", None), "fnum": getattr(self, "fnum", None),
                "r_last": float(self.r_last), "d_sensor": float(self.d_sensor),
                "sensor_size": list(self.sensor_size), "surfaces": []}
        for i, s in enumerate(self.surfaces):
            sd = s.surf_dict()
            nxt = self.surfaces[i + 1].d if i < len(self.surfaces) - 1 else self.d_sensor
            sd["d_next"] = float(nxt) - float(s.d)
            data["surfaces"].append(sd)
        with open(filename, "w") as f:
            json.dump(data, f, indent=4)

    # nn.Module-style switches: the reference's Lensgroup / PSFNet inherit them from DeepObj(nn.Module)
    # (basics.py:165-213) and its scripts call them (dfdp/factory.py:15,31-32: lens.to(device), lens.eval())
    _DEVICE_CACHES = ("_stage_ring", "_sample_stream", "_readback_stream", "_ctl_pools", "_p2o_cache", "_ctl_host", "_ctl_host_ring",
                      "_right_streak", "_n_cus", "_pinned_out")

    def train(self, mode=True):
        """The PSF network (if this lens carries one) in training / evaluation mode; -> self."""
        net = getattr(self, "psfnet", None)
        if net is not None:
            net.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def to(self, device):
        """basics.py:180-201: use `device` from now on.  The device-side lens tables, streams and staging buffers
        belong to the old device and are dropped (rebuilt on first use); the PSF network moves with the lens."""
        device = _as_device(device)
        if device != self.device:
            self.device = device
            self._invalidate()
            for name in list(self.__dict__):
                if name in self._DEVICE_CACHES or name.endswith("_stream"):
                    del self.__dict__[name]
            self.last_pupil_points = None
        net = getattr(self, "psfnet", None)
        if net is not None:
            net.to(device)
            if hasattr(net, "invalidate_packed"):
                net.invalidate_packed()
        return self

    def _invalidate(self):
        self._dev.clear()
        self._pupil_cache.clear()
        self.__dict__.pop("_curved_cache", None)

    def find_aperture(self):
        """optics.py:193-201: first surface with air-like media on both sides."""
        self.aper_idx = None
        for i in range(len(self.surfaces) - 1):
            if self.surfaces[i].mat1.A < 1.0003 and self.surfaces[i].mat2.A < 1.0003:
                self.aper_idx = i
                return

    def find_diff_surf(self):
        if self.aper_idx is None:
            return range(len(self.surfaces))
        return list(range(0, self.aper_idx)) + list(range(self.aper_idx + 1, len(self.surfaces)))

    def prepare_sensor(self, sensor_res=(512, 512), sensor_size=(24.0, 36.0)):
        """optics.py:154-178."""
        sensor_res = [sensor_res, sensor_res] if isinstance(sensor_res, int) else list(sensor_res)
        self.sensor_res = sensor_res
        H, W = sensor_res
        if sensor_size is None:
            self.sensor_size = [2 * self.r_last * H / np.sqrt(H ** 2 + W ** 2),
                                2 * self.r_last * W / np.sqrt(H ** 2 + W ** 2)]
        else:
            self.sensor_size = list(sensor_size)
            self.r_last = np.sqrt(sensor_size[0] ** 2 + sensor_size[1] ** 2) / 2
        assert self.sensor_size[0] / self.sensor_size[1] == H / W, "Pixel is not square."
        self.pixel_size = self.sensor_size[0] / sensor_res[0]

    def post_computation(self):
        """optics.py:181-190."""
        self.find_aperture()
        self.hfov = self.calc_fov()
        self.foclen = self.calc_efl()
        _, avg_pupilx = self.entrance_pupil()
        self.fnum = self.foclen / avg_pupilx / 2

    def set_state(self, d_sensor=None, hfov=None, pupil=None, exit_pupil=None):
        """Pin the geometric-optics scalars (e.g. from a reference fixture) instead
        of computing them; the reference's own values drift run to run
        (its paraxial pupil is an ill-conditioned fp32 lstsq, optics.py:1500)."""
        if d_sensor is not None:
            self.d_sensor = float(d_sensor)
        if hfov is not None:
            self.hfov = float(hfov)
            self.foclen = self.calc_efl()
        if pupil is not None:
            self._pupil_cache[True] = (float(pupil[0]), float(pupil[1]))
        if exit_pupil is not None:
            self._pupil_cache[False] = (float(exit_pupil[0]), float(exit_pupil[1]))
        if pupil is not None and hasattr(self, "foclen"):
            self.fnum = self.foclen / self._pupil_cache[True][1] / 2
        return self

    # ----------------------------------------------------------- device side
    def _require_gpu(self):
        if self.device.type != "cuda":
            raise _lib.SdirtError("sdirt_amd traces rays on the GPU only; construct the lens with "
                                  "device='cuda' (no CPU fallback exists)")

    def dev_lens(self, wvln=DEFAULT_WAVE):
        self._require_gpu()
        key = float(wvln if wvln < 10 else wvln * 1e-3)
        dl = self._dev.get(key)
        if dl is None:
            with torch.cuda.device(self.device):
                dl = _DevLens(self.surfaces, key)
            # the handle is only cached -- and handed out -- once the library has passed its self-test on this
            # prescription: a failed test raises on THIS and on every later call (the failure is remembered per
            # device and prescription, _selftest_failed), never just on the first
            self._prefetch_selftest(dl.handle)
            self._dev[key] = dl
        return dl.handle

    def _prefetch_selftest(self, handle):
        """Second guard of the hand-scheduled scalar prefetch in the trace loop (the first is the
        ISA check of the build, tools/check_prefetch_hazard.py): when a prescription is first
        uploaded, a fan of probe rays is traced twice -- next surface's constants prefetched one
        surface ahead / loaded on the spot (SDIRT_TRACE_NO_PREFETCH) -- forward and backward, both
        math policies; the two must agree bit for bit.  A compiler that moved an instruction into
        the in-flight window would make the prefetching loop trace with stale constants or trip
        counts; then every call on this lens raises instead of returning wrong PSFs.  Draws no random numbers."""
        digest = self._table_digest()
        if Lensgroup._selftest_done.get((self.device.index, len(self.surfaces))) == digest:
            return
        if (self.device.index, digest) in Lensgroup._selftest_failed:
            raise _lib.SdirtError(Lensgroup._selftest_failed[(self.device.index, digest)])
        K = len(self.surfaces)
        front, back = self.surfaces[0], self.surfaces[-1]
        n = 16
        lin = torch.linspace(-0.6, 0.6, n, device=self.device)
        gx, gy = torch.meshgrid(lin, lin, indexing="xy")
        trips = (C.c_int32 * K)(*self._fixed_trips_for("max").tolist())
        for forward in (True, False):
            surf = front if forward else back
            z0 = float(surf.d) - 40.0 if forward else float(self.d_sensor)
            o = torch.stack((gx * 3.0, gy * 3.0 + 0.5, torch.full_like(gx, z0)), -1).reshape(-1, 3)
            aim = torch.stack((gx * float(surf.r), gy.flip(0) * float(surf.r),
                               torch.full_like(gx, float(surf.d))), -1).reshape(-1, 3)
            for flags in (0, _lib.PSF_STRICT_IEEE):
                a = Ray(o, aim - o, device=self.device)
                b = a.clone()
                for ray, extra in ((a, 0), (b, _lib.TRACE_NO_PREFETCH)):
                    _lib.check(_lib.lib().sdirt_trace(handle, 0, K, 0 if forward else 1, trips, flags | extra,
                                                      ray.c_rays(), ray.numel, None, stream_ptr(self.device)))
                if not torch.equal(a.soa.view(torch.int32), b.soa.view(torch.int32)):
                    msg = ("libsdirt_dp.so self-test failed: the trace loop with prefetched surface constants "
                           "disagrees with the load-and-wait form (miscompiled surf_issue/surf_wait window?); "
                           "rebuild with `make -C sdirt_amd/csrc` and check tools/check_prefetch_hazard.py")
                    Lensgroup._selftest_failed[(self.device.index, digest)] = msg
                    raise _lib.SdirtError(msg)
        Lensgroup._selftest_done[(self.device.index, K)] = digest

    _selftest_done = {}
    _selftest_failed = {}          # (device index, prescription digest) -> message: sticky

    def _table_digest(self):
        return hash(tuple((s.kind, float(s.r), float(s.d), float(s.c), float(s.k),

                           tuple(float(a) for a in (s.ai if s.ai is not None else ()))) for s in self.surfaces))

    def _fixed_trips_for(self, policy):
        n = NEWTON_MAXITER if policy == "max" else -NEWTON_MAXITER
        return np.where(self._curved(), n, 0).astype(np.int32)

    def _math_flags(self):
        if self.precision not in ("ieee", "lean"):
            raise ValueError("precision must be 'lean' or 'ieee'")
        return _lib.PSF_STRICT_IEEE if self.precision == "ieee" else 0

    def _psf_flags(self):
        """Math policy and, on request, the order-independent sums of a fused psf call (self.deterministic)."""
        return self._math_flags() | (_lib.PSF_DETERMINISTIC if self.deterministic else 0)

    def _timed(self, name):
        return _EventBracket(self.kernel_events, name, self.device)

    def _curved(self):
        c = self.__dict__.get("_curved_cache")
        if c is None or len(c) != len(self.surfaces):
            c = self.__dict__["_curved_cache"] = [s.kind != _lib.KIND_PLANE for s in self.surfaces]
        return c

    def _mask_buffer(self):
        return torch.zeros(_lib.MAX_SURFACES, dtype=torch.int32, device=self.device)

    def _read_masks(self, mask):
        m = mask[:len(self.surfaces)]
        if self.mask_reduce is not None:
            m = self.mask_reduce(m)
        return m.cpu().numpy().astype(np.int64) & 0xFFFFFFFF

    def _fixed_trips(self):
        """Trip table of the policies that need no host check: 10 everywhere ('max') or -10 =
        "up to 10, each wave stops when its own rays are done" ('adaptive')."""
        if self.trip_policy not in ("max", "adaptive"):
            raise ValueError(f"unknown trip_policy {self.trip_policy!r}")
        return self._fixed_trips_for(self.trip_policy)

    def _run_with_trips(self, key, order, enqueue):
        """enqueue(trips_ctypes, mask_ptr) launches the kernels.  Returns the trip
        table that was finally used."""
        K = len(self.surfaces)
        curved = self._curved()
        if self.trip_policy != "reference":
            trips = self._fixed_trips()
            enqueue((C.c_int32 * K)(*trips.tolist()), None)
            return trips
        mask = self._mask_buffer()

        def launch(trips):
            mask.zero_()
            enqueue((C.c_int32 * K)(*[int(t) for t in trips]), dptr(mask))
            return self._read_masks(mask)

        return self.trips.run(key, curved, list(order), launch)

    # ---------------------------------------------------------------- sampling
    @torch.no_grad()
    def sample_from_points(self, o=[[0, 0, -10000]], spp=256, wvln=DEFAULT_WAVE,
                           shrink_pupil=False, normalized=False):
        """optics.py:460-494: rays [spp, N] from object-space points to random pupil points."""
        self._require_gpu()
        if not torch.is_tensor(o):
            o = torch.tensor(o)
        po = o.to(self.device, torch.float32).reshape(-1, 3).contiguous()
        pupilz, pupilr = self.entrance_pupil(shrink_pupil=shrink_pupil)
        x2, y2 = self._pupil_samples(spp, pupilr)
        ray = Ray.empty((spp, po.shape[0]), wvln, self.device)
        _lib.check(_lib.lib().sdirt_sample_rays(dptr(po), po.shape[0], dptr(x2), dptr(y2), spp,
                                                float(pupilz), ray.c_rays(),
                                                stream_ptr(self.device)))
        if basics.TRACK_OBLIQ:                               # a caller's bundle: readable obliq, as the reference's (basics.py:240)
            ray.obliq = torch.ones((), device=self.device)
        return ray

    def _staging(self, spp, depth=8, rows=2):
        """A page-locked [rows, spp] buffer from a ring of `depth` per size, and the event that
        marks its upload as done; a slot is reused only after its previous upload has finished."""
        rings = self.__dict__.setdefault("_stage_ring", {})
        if (rows, spp) not in rings and len(rings) >= 16:      # a caller sweeping over many sizes: keep the newest rings
            rings.pop(next(iter(rings)))
        ring = rings.setdefault((rows, spp), {"slots": [], "next": 0})
        if len(ring["slots"]) < depth:
            ring["slots"].append((torch.empty((rows, spp), dtype=torch.float32, pin_memory=True),
                                  torch.cuda.Event()))
            return ring["slots"][-1]
        slot = ring["slots"][ring["next"]]
        ring["next"] = (ring["next"] + 1) % depth
        slot[1].synchronize()
        return slot

    def _pupil_samples(self, spp, pupil_r):
        """optics.py:483-488: spp points on the pupil disc -> device arrays (x2, y2).

        The two uniform vectors always come from torch's CPU default generator in the
        reference's order (that is what makes seeds line up).  pupil_mapping='device'
        (default) maps them to the disc with the sdirt_pupil_samples kernel (correctly rounded
        sin/cos).  'host' evaluates the reference's own CPU tensor expressions
        (optics.py:483-486) instead: bit-identical sample points to the reference ON THE SAME
        MACHINE -- torch's sin/cos are MKL kernels whose last bit differs between CPU models
        (the Xeon that produced tests/golden and the EPYC of the GPU box disagree on ~5 % of
        the samples) -- at the price of threaded MKL calls (measured 24 ms per call on a
        256-thread host under a 16-CPU quota)."""
        if self.pupil_mapping == "host":
            u_theta, u_r2 = torch.rand(spp), torch.rand(spp)
            theta = u_theta * 2 * np.pi
            r = torch.sqrt(u_r2 * pupil_r ** 2)
            xy = torch.stack((r * torch.cos(theta), r * torch.sin(theta))).to(self.device)
            return xy[0], xy[1]
        # draw straight into page-locked memory (same generator, same values as torch.rand(spp))
        # so that the upload is a true asynchronous copy: from pageable memory it would block
        # the host until the stream has drained, i.e. until the previous call's kernel is done.
        # Upload and mapping depend on nothing the caller has queued, so they go on a side stream
        # (they run beside the previous call's kernel instead of behind it) and the caller's
        # stream waits for their event.
        main = torch.cuda.current_stream(self.device)
        side = self.__dict__.get("_sample_stream")
        if side is None:
            side = self.__dict__["_sample_stream"] = torch.cuda.Stream(self.device)
        with torch.cuda.stream(side):
            stage, done = self._staging(spp)
            _hostrng.rand_into(stage[0])
            _hostrng.rand_into(stage[1])
            u = stage.to(self.device, non_blocking=True)
            done.record(side)
            xy = torch.empty((2, spp), dtype=torch.float32, device=self.device)
            _lib.check(_lib.lib().sdirt_pupil_samples(dptr(u[0]), dptr(u[1]), spp, float(pupil_r),
                                                      dptr(xy[0]), dptr(xy[1]),
                                                      stream_ptr(self.device)))
        main.wait_stream(side)
        xy.record_stream(main)
        return xy[0], xy[1]

    def _pupil_samples_pair(self, spp, pupil_r, spp_c, pupil_r_c, side_stream=True):
        """The two sample sets of one psf call -- `spp` points on the pupil, then `spp_c` on the
        shrunk pupil of the chief-ray pass -- with ONE draw, ONE upload and one device block:
        torch.rand(2 spp + 2 spp_c) is, value for value, the four consecutive draws
        rand(spp), rand(spp), rand(spp_c), rand(spp_c) of the reference (optics.py:483-484 twice;
        the CPU generator hands out one 24-bit number per element, whatever the call sizes:
        tests/test_host_logic.py).  -> (x2, y2, xc, yc)."""
        if self.pupil_mapping == "host":
            return self._pupil_samples(spp, pupil_r) + self._pupil_samples(spp_c, pupil_r_c)
        n = 2 * (spp + spp_c)
        a, b = 2 * spp, 2 * spp + spp_c
        h = _lib.lib()

        def upload_and_map():
            stage, done = self._staging(n, rows=1)
            _hostrng.rand_into(stage[0])
            u = stage[0].to(self.device, non_blocking=True)
            done.record(torch.cuda.current_stream(self.device))
            xy = torch.empty(n, dtype=torch.float32, device=self.device)
            st = stream_ptr(self.device)
            pu, pxy = u.data_ptr(), xy.data_ptr()
            P = lambda base, off: C.c_void_p(base + 4 * off)
            _lib.check(h.sdirt_pupil_samples(P(pu, 0), P(pu, spp), spp, float(pupil_r), P(pxy, 0), P(pxy, spp), st))
            _lib.check(h.sdirt_pupil_samples(P(pu, a), P(pu, b), spp_c, float(pupil_r_c), P(pxy, a), P(pxy, b), st))
            return u, xy

        if not side_stream:
            # synchronous caller: nothing of an earlier call is in flight that the upload could run
            # beside -- straight onto the caller's stream, no stream switch, no cross-stream waits
            u, xy = upload_and_map()
        else:
            main = torch.cuda.current_stream(self.device)
            side = self._side_stream("_sample_stream")
            with torch.cuda.stream(side):
                u, xy = upload_and_map()
            main.wait_stream(side)
            xy.record_stream(main)
        return xy[:spp], xy[spp:a], xy[a:b], xy[b:]

    # ----------------------------------------------------------------- tracing
    def trace(self, ray, lens_range=None, record=False, forward=None, _to_sensor=None):
        """optics.py:601-627: updates `ray` and returns (ray, valid, oss).  Direction is
        taken from the first ray's d_z like the reference unless `forward` is given.
        Under trip_policy 'reference' the trace runs OUT OF PLACE and `ray`'s storage is rebound to the traced
        bundle (the reference rebinds ray.o / ray.d / ray.ra to new tensors too, surfaces.py:425, 676-677): views
        (`ray.ra`, `ray.obliq`) and `c_rays()` pointers taken BEFORE the call keep showing the untraced rays --
        re-fetch them after trace / trace2sensor.  ('max' / 'adaptive' trace in place.)
        _to_sensor = z: trace2sensor's form -- forward through all surfaces and on to the plane z in the same pass
        (sdirt_trace2sensor)."""
        self._require_gpu()
        K = len(self.surfaces)
        if record:
            return self._trace_recorded(ray, lens_range, forward)
        if lens_range is None:
            first, last = 0, K
        else:
            lr = list(lens_range)
            first, last = (lr[0], lr[-1] + 1) if lr else (0, 0)
            assert lr == list(range(first, last)), "lens_range must be contiguous"
        if forward is None:
            forward = bool(ray.soa[5, 0].item() > 0)             # optics.py:618
        order = list(range(first, last)) if forward else list(range(last - 1, first - 1, -1))
        handle = self.dev_lens(ray.wvln)

        def enqueue(trips, mask_ptr):
            _lib.check(_lib.lib().sdirt_trace(handle, first, last, 0 if forward else 1, trips,
                                              self._math_flags(), ray.c_rays(), ray.numel,
                                              mask_ptr, stream_ptr(self.device)))

        if self.trip_policy == "reference":
            # a wrong bet on the trip table means tracing the batch again from its input: the trace goes OUT OF PLACE
            # into a second bundle (sdirt_trace_to) which then becomes the ray's storage -- the reference's trace
            # rebinds ray.o / ray.d / ray.ra to new tensors in the same way (surfaces.py:425, 676-677); copying the
            # bundle first to trace in place cost as much memory traffic as the trace itself
            dst = Ray.empty(ray.shape, ray.wvln, self.device, obliq=ray.has_obliq)

            def enqueue_to(trips, mask_ptr):
                if _to_sensor is not None:
                    _lib.check(_lib.lib().sdirt_trace2sensor(handle, trips, self._math_flags(), float(_to_sensor),
                                                             ray.c_rays(), dst.c_rays(), ray.numel, mask_ptr,
                                                             stream_ptr(self.device)))
                    return
                _lib.check(_lib.lib().sdirt_trace_to(handle, first, last, 0 if forward else 1, trips,
                                                     self._math_flags(), ray.c_rays(), dst.c_rays(), ray.numel,
                                                     mask_ptr, stream_ptr(self.device)))
            key = ("trace", round(float(ray.wvln), 6), first, last, forward, self.precision)
            self._run_with_trips(key, order, enqueue_to)
            ray._adopt(dst)
        else:
            self._run_with_trips(None, order, enqueue)
            ray._mark_traced()
            if _to_sensor is not None:
                ray.propagate_to(_to_sensor)
        valid = ray.ra == 1
        return ray, valid, None

    def _trace_recorded(self, ray, lens_range, forward):
        """trace(record=True), optics.py:666-717: `oss[i]` = the points ray i (row i of the first batch
        dimension) passed through -- its origin, then its position after every surface it left alive --
        as numpy arrays, for the reference's ray-path plots.  Traced surface by surface with the
        staged kernel (each surface's batch-wide Newton trip count verified on its own), the positions
        copied to the host after every surface: a plotting aid, not a fast path."""
        K = len(self.surfaces)
        idx = list(range(K)) if lens_range is None else list(lens_range)
        if forward is None:
            forward = bool(ray.soa[5, 0].item() > 0)             # optics.py:618
        oss = [[p] for p in ray.o.cpu().numpy()]
        for k in (idx if forward else idx[::-1]):
            self.trace(ray, lens_range=range(k, k + 1), forward=forward)
            alive = ((ray.ra == 1) if forward else (ray.ra > 0)).cpu().numpy()   # optics.py:681 / :707
            for path, v, p in zip(oss, alive, ray.o.cpu().numpy()):
                if np.any(v):
                    path.append(p)
        return ray, ray.ra == 1, oss

    def trace2sensor(self, ray, record=False, ignore_invalid=False):
        """optics.py:638-664 (the ray's storage is rebound like trace()'s).  record=True: (p, oss) -- the sensor-plane positions [M,3] and the
        recorded paths, each live ray's sensor point appended (twice, as the reference's two loops do)."""
        if not record:
            if bool(ray.soa[5, 0].item() > 0):                   # forward (optics.py:618): one pass, sdirt_trace2sensor
                ray, _, _ = self.trace(ray, forward=True, _to_sensor=self.d_sensor)
                return ray
            ray, _, _ = self.trace(ray)
            return ray.propagate_to(self.d_sensor)
        ray, _, oss = self.trace(ray, record=True)
        ray.propagate_to(self.d_sensor)
        valid, p = ray.ra == 1, ray.o
        for path, v, pp in zip(oss, valid.cpu().numpy(), p.cpu().numpy()):
            if np.any(v):
                path.append(pp)
        if ignore_invalid:
            p = p[valid]
        else:
            assert p.dim() >= 2, "This function is not tested."
            p = p.reshape(-1, 3)
        for v, path, pp in zip(valid.cpu().numpy(), oss, p.cpu().numpy()):
            if v:
                path.append(pp)
        return p, oss

    # --------------------------------------------------------------------- PSF
    def point_source_grid(self, depth, grid=9, normalized=True, quater=False, center=False):
        """optics.py:816-861: [grid, grid, 3] field of point sources at one depth; entry [i, j] is
        (x_j, y_i, depth) with x running left to right and y from the top edge down.  Edge samples
        sit at +-0.98 of the normalised field (or at patch centres, +-(1 - 1/(2(grid-1))), with
        center=True); quater keeps the upper-left ceil(grid/2) x ceil(grid/2) block."""
        if grid == 1:
            assert not quater, "Quater should be False when grid is 1."
            axis = torch.zeros(1)
        else:
            edge = 1 - 1 / 2 / (grid - 1) if center else 0.98
            axis = torch.linspace(-edge, edge, grid)
        field = torch.empty(grid, grid, 3)
        field[..., 0] = axis.view(1, grid)
        field[..., 1] = -axis.view(grid, 1)
        field[..., 2] = float(depth)
        if quater:
            keep = (grid + 1) // 2
            field = field[:keep, :keep]
        if not normalized:
            # object-space millimetres; x takes the sensor HEIGHT here (the reference's convention in
            # this function, optics.py:858-859 -- psf_diff uses the width for x)
            scale = self.calc_scale_pinhole(depth)
            field[..., 0] *= scale * self.sensor_size[0] / 2
            field[..., 1] *= scale * self.sensor_size[1] / 2
        return field

    def _side_stream(self, name):
        st = self.__dict__.get(name)
        if st is None:
            st = self.__dict__[name] = torch.cuda.Stream(self.device)
        return st

    def _zeroed_control_block(self, n, rows=64):
        """An int32 [n] device block that is zero, cut from a pool that is cleared `rows` blocks
        at a time (one fill kernel per `rows` launches instead of one per launch)."""
        # one pool per (size, stream): the fill runs on the stream that is current when the pool is
        # created, and rows are only ever handed to launches on THAT stream -- so the fill is ordered
        # before every kernel that ORs into a row, and the caching allocator (which hands memory back
        # to the stream it was allocated on) cannot recycle the pool under a foreign stream's kernel
        pools = self.__dict__.setdefault("_ctl_pools", {})
        key = (n, torch.cuda.current_stream(self.device).cuda_stream)
        pool = pools.get(key)
        if pool is None or pool["next"] >= rows:
            pools.pop(key, None)
            while len(pools) >= 8:                     # a caller sweeping over many shapes: keep the newest pools only
                pools.pop(next(iter(pools)))
            pool = pools[key] = {"next": 0,
                                 "buf": torch.zeros((rows, n), dtype=torch.int32, device=self.device)}
        row = pool["buf"][pool["next"]]
        pool["next"] += 1
        return row

    def _points_to_object(self, points):
        """[N,3] normalised points -> object space (optics.py:959-960, 1305), on the device.  The
        result for the LAST tensor is kept: a caller that renders the same grid call after call (a
        PSF volume, an evaluation set) pays for the conversion once.  The key is the tensor object
        itself, its version counter (any in-place write bumps it) and every lens scalar used."""
        # only device tensors that track a version counter are cached: a CPU tensor can be written
        # behind torch's back (torch.from_numpy + a numpy-side store leaves _version unchanged), an
        # inference-mode tensor has no counter at all
        if not points.is_cuda or points.is_inference():
            return self._points_to_object_now(points)
        key = (points._version, tuple(points.shape), points.dtype, float(np.tan(self.hfov)),
               float(self.r_last), float(self.sensor_size[1]), float(self.sensor_size[0]),
               str(self.device), torch.cuda.current_stream(self.device).cuda_stream)
        # one entry per stream (a caller that alternates two render streams keeps both): the converted points were
        # produced on that stream and are only handed to launches on it
        cache = self.__dict__.setdefault("_p2o_cache", {})
        hit = cache.get(key[-1])
        if hit is not None and hit[0]() is points and hit[1] == key:
            return hit[2]
        out = self._points_to_object_now(points)
        if key[-1] not in cache and len(cache) >= 4:
            cache.pop(next(iter(cache)))
        cache[key[-1]] = (weakref.ref(points), key, out)
        return out

    def _points_to_object_now(self, points):
        if not points.is_cuda:
            # page-locked staging: an upload from pageable memory blocks the host until the
            # stream has drained (i.e. until the previous call's kernel has finished).  The buffer comes from
            # the lens's ring of page-locked blocks (Tensor.pin_memory() would page-lock a fresh allocation on
            # every call: a system call of its own, ~0.3 ms for the 24576 points of the reference's timing harness)
            n = points.shape[0]
            stage, uploaded = self._staging(max(n, 1), depth=4, rows=3)
            host = stage.view(-1)[:3 * n].view(n, 3)
            if points.dtype == torch.float32 and points.is_contiguous() and not points.requires_grad:
                # a plain memcpy: torch's CPU copy_ is an OpenMP-parallel op whose worker threads spin afterwards --
                # inside a CPU-quota'd container that can cost the calling thread its time slice (tools/tcp_span.py)
                np.copyto(host.numpy(), points.numpy())
            else:
                host.copy_(points)
            pts = host.to(self.device, non_blocking=True)
            uploaded.record(torch.cuda.current_stream(self.device))
        else:
            pts = points.to(self.device, torch.float32, non_blocking=True).contiguous()
        out = torch.empty_like(pts)
        _lib.check(_lib.lib().sdirt_points_to_object(
            dptr(pts), pts.shape[0], float(np.tan(self.hfov)), float(self.r_last),
            float(self.sensor_size[1]), float(self.sensor_size[0]), dptr(out),
            stream_ptr(self.device)))
        return out

    @torch.no_grad()
    def psf_center(self, point, method="chief_ray"):
        """optics.py:889-914: [N,3] object-space points -> [N,2] PSF centres (green light)."""
        if method == "pinhole":
            scale = self.calc_scale_pinhole(point[..., 2])
            return -point[..., :2] / scale
        if method != "chief_ray":
            raise Exception("Unsupported method.")
        self._require_gpu()
        po = point.to(self.device, torch.float32).reshape(-1, 3).contiguous()
        pupilz, pupilr = self.entrance_pupil(shrink_pupil=True)
        xc, yc = self._pupil_samples(GEO_SPP, pupilr)
        center = torch.empty((po.shape[0], 2), dtype=torch.float32, device=self.device)
        self._chief_center(po, xc, yc, pupilz, center)
        return center

    def _chief_center(self, po, xc, yc, pupilz, center):
        anyv = torch.zeros(1, dtype=torch.int32, device=self.device)
        handle = self.dev_lens(DEFAULT_WAVE)                      # optics.py:900: always green

        def enqueue(trips, mask_ptr):
            with self._timed("chief_center"):
                _lib.check(_lib.lib().sdirt_chief_center(
                    handle, dptr(po), po.shape[0], dptr(xc), dptr(yc), xc.shape[0], float(pupilz),
                    float(self.d_sensor), trips, self._math_flags(), dptr(center), dptr(anyv),
                    mask_ptr, stream_ptr(self.device)))
        self._run_with_trips(("center", self.precision), range(len(self.surfaces)), enqueue)
        if self.trip_policy == "reference":
            assert int(anyv.item()) == 1, "No sampled rays is valid."   # optics.py:902
        return center

    def psf(self, points, ks=31, wvln=DEFAULT_WAVE, spp=GEO_SPP, center=True):
        """optics.py:916-931."""
        return self.psf_diff(points=points, wvln=wvln, ks=ks, spp=spp, center=center)

    def psf_diff(self, points, wvln=DEFAULT_WAVE, ks=31, spp=GEO_SPP, center=True,
                 param_list=None, _defer=False):
        """optics.py:934-996: normalised points [N,3] (or [3]) -> max-normalised
        PSF [N,ks,ks] (or [ks,ks]) of the left sub-pixel (right if param_list[4] != 'l')."""
        dp, direct = None, "l"
        if param_list is not None:
            h, f, w, r, direct = param_list
            dp = (h, f, w, r)
        res = self.psf_lr(points, ks=ks, wvln=wvln, spp=spp, center=center, dp=dp,
                          want_r=(param_list is not None and direct != "l"),
                          _default_r_zero=(param_list is None), defer=_defer)
        pick = (lambda lr: lr[0]) if direct == "l" else (lambda lr: lr[1])
        if _defer:
            return PendingPSF(lambda: pick(res.wait()))
        return pick(res)

    def to_host(self, t):
        """`t.to('cpu')` at PCIe speed: the copy goes into a PAGE-LOCKED buffer kept on the lens (one per shape and
        dtype, the two most recent kept) and the caller's stream is waited for -- a pageable destination, what
        Tensor.to('cpu') allocates, moved the 42.5 MB of the reference's timing harness (psfnet.py:570-586) at ~7 GB/s,
        a third of its span.  The returned CPU tensor IS that buffer: valid until the next to_host() of the same shape;
        `.clone()` it to keep it."""
        if not t.is_cuda:
            return t
        pool = self.__dict__.setdefault("_pinned_out", {})
        key = (tuple(t.shape), t.dtype)
        host = pool.pop(key, None)
        if host is None:
            while len(pool) >= 2:
                pool.pop(next(iter(pool)))
            host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        pool[key] = host                                    # most recent last
        host.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
        return host

    def _spp_slices(self, N, spp):
        """sdirt_psf_spp_slices for THIS lens's GPU (its CU count, whatever the caller's current device is)."""
        ncu = self.__dict__.get("_n_cus")
        if ncu is None:
            ncu = self.__dict__["_n_cus"] = int(torch.cuda.get_device_properties(self.device).multi_processor_count)
        return _lib.lib().sdirt_psf_spp_slices(N, spp, ncu)

    def _centre_buffer(self, center_out, N):
        """The [N, 2] centre tensor of a psf call: the caller's (checked) or a fresh one."""
        if center_out is None:
            return torch.empty((N, 2), dtype=torch.float32, device=self.device)
        if not (center_out.is_cuda and center_out.dtype == torch.float32 and center_out.is_contiguous()
                and tuple(center_out.shape) == (N, 2)):
            raise ValueError("center_out must be a contiguous float32 CUDA [N, 2] tensor")
        return center_out

    def _psf_buffers(self, out, N, ks, need_r):
        """(L, R) [N, ks, ks] of a psf call: the caller's `out` pair (checked) or fresh tensors; R is None when
        the call fills no right grid.  out = ONE float32 CUDA [N, 2, ks, ks] tensor: L = out[:, 0], R = out[:, 1]
        (SDIRT_PSF_INTERLEAVED: the block a multi-GPU volume gathers with one collective, dist.py)."""
        if torch.is_tensor(out):
            if not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
                    and tuple(out.shape) == (N, 2, ks, ks)):
                raise ValueError("out as ONE tensor must be contiguous float32 CUDA [N, 2, ks, ks]")
            if not need_r:
                raise ValueError("out=[N, 2, ks, ks] holds a left AND a right grid per point: needs dp and want_r")
            return out[:, 0], out[:, 1]
        if out is None:
            L = torch.empty((N, ks, ks), dtype=torch.float32, device=self.device)
            return L, (torch.empty_like(L) if need_r else None)
        L, R = out[0], (out[1] if need_r else None)
        for t_ in (L, R):
            if t_ is not None and not (t_.is_cuda and t_.dtype == torch.float32 and t_.is_contiguous()
                                       and tuple(t_.shape) == (N, ks, ks)):
                raise ValueError("out tensors must be contiguous float32 CUDA [N, ks, ks]")
        return L, R

    @torch.no_grad()
    def psf_lr(self, points, ks=31, wvln=DEFAULT_WAVE, spp=GEO_SPP, center=True,
               dp=(0.78, 1.44, 0.3, 0.5), normalize=True, want_r=True, _default_r_zero=False,
               pupil_xy=None, center_pupil_xy=None, out=None, defer=False, center_out=None):
        """Left AND right dual-pixel PSFs of one ray-traced batch: (L, R), each
        [N,ks,ks] (or [ks,ks] for a single point), max-normalised separately as
        optics.py:983-987 would normalise each of them.  dp = (h, f, w, r) of
        monte_carlo.py:157-164.

        pupil_xy / center_pupil_xy: optional explicit pupil sample points
        (x2[spp], y2[spp]) for the primary and the chief-ray pass; when given,
        no random numbers are drawn for that pass (used for ray-level parity
        hand-off and for quasi-random sampling).

        out: optional (L, R) float32 CUDA tensors [N,ks,ks] to write into -- a consumer
        that renders batch after batch (PSFNet fitting) re-uses its buffers instead of
        asking the caching allocator for two 277 MB blocks per call.  Or ONE [N,2,ks,ks]
        tensor: the returned L and R are its views out[:, 0] and out[:, 1] (a rank of a
        sharded volume renders straight into the block that one all-gather moves).

        center_out: optional float32 CUDA [N,2] tensor that receives the PSF centres the splat used
        (the chief-ray centres of optics.py:969 with center=True).

        defer=True (center=True only): enqueue the kernel with the speculated Newton trip tables
        and return a `PendingPSF` at once; its `.wait()` reads the convergence masks back,
        verifies them (re-launching in the rare case the speculation was wrong) and returns
        (L, R).  A caller that renders batch after batch keeps one call in flight -- the GPU
        starts batch i+1 while the host checks batch i -- without giving up the check."""
        self._require_gpu()
        if not torch.is_tensor(points):
            points = torch.tensor(points)
        single_point = points.dim() == 1
        if single_point:
            points = points.unsqueeze(0)
        N = points.shape[0]
        if torch.is_tensor(out) and (dp is None or _default_r_zero or not want_r):
            raise ValueError("out=[N, 2, ks, ks] holds a left AND a right grid per point: needs dp and want_r")
        if N == 0 and not (self.mask_reduce is not None and center
                           and self.trip_policy == "reference"):
            # empty batch: nothing to trace, no random numbers drawn
            e = torch.empty((0, ks, ks), dtype=torch.float32, device=self.device)
            res = (e, (e.clone() if want_r else None))
            return PendingPSF(lambda: res) if defer else res
        # (an empty SHARD of a multi-rank call goes on: its rank must enter the same mask
        # reductions and take the same re-launch decisions as its peers, with nothing to launch)
        po = self._points_to_object(points) if N else torch.empty((0, 3), device=self.device)
        # RNG order of the reference: primary pupil samples first (optics.py:963),
        # then the chief-ray samples inside psf_center (optics.py:969).
        pupilz, pupilr = self.entrance_pupil()
        if ks > _lib.MAX_KS:
            # a point's two grids no longer fit in LDS (draw_mtf: ks 256): the staged chain
            if defer or torch.is_tensor(out):
                raise ValueError(f"defer=True / out=[N, 2, ks, ks] need ks <= {_lib.MAX_KS}")
            return self._psf_lr_staged(points, po, N, ks, wvln, spp, center, dp, normalize, want_r,
                                       _default_r_zero, pupil_xy, center_pupil_xy, out, center_out, single_point)
        if (center and pupil_xy is None and center_pupil_xy is None and N > 0 and (not defer or self.defer_one_call)
                and self.trip_policy == "reference" and self.mask_reduce is None and self.pupil_mapping == "device"
                and self.kernel_events is None and self._spp_slices(N, spp) > 1):
            # the fitting shape (few points, many samples): launch-latency-bound, so everything between the random
            # draw and the check of what the device did is ONE library call -- followed by a stream synchronisation
            # (the synchronous call) or by a PendingPSF that waits for the call's own event (defer=True: the fitting
            # loop keeps two batches in flight; ~0.1 ms of host work per batch instead of ~0.4 through the general path)
            return self._psf_call_one(points, po, N, ks, wvln, spp, dp, normalize, want_r, _default_r_zero, out,
                                      center_out, single_point, defer=defer)
        both_drawn = None
        if pupil_xy is None and center and center_pupil_xy is None:
            # both sample sets of the call in one draw / upload (same numbers, same order)
            both_drawn = self._pupil_samples_pair(spp, pupilr, GEO_SPP, self.entrance_pupil(shrink_pupil=True)[1],
                                                  side_stream=defer)
            x2, y2 = both_drawn[:2]
        elif pupil_xy is None:
            x2, y2 = self._pupil_samples(spp, pupilr)
        else:
            x2, y2 = [torch.as_tensor(v).to(self.device, torch.float32).contiguous()
                      for v in pupil_xy]
            spp = x2.shape[0]
        cen = self._centre_buffer(center_out, N)
        xc = yc = None
        if center:
            _, pupilr_c = self.entrance_pupil(shrink_pupil=True)
            if both_drawn is not None:
                xc, yc = both_drawn[2:]
            elif center_pupil_xy is None:
                xc, yc = self._pupil_samples(GEO_SPP, pupilr_c)
            else:
                xc, yc = [torch.as_tensor(v).to(self.device, torch.float32).contiguous()
                          for v in center_pupil_xy]
        else:
            pts = points.to(self.device, torch.float32)
            cen[:, 0] = pts[:, 0] * (self.sensor_size[1] / 2)      # optics.py:973-975
            cen[:, 1] = pts[:, 1] * (self.sensor_size[0] / 2)
        #: the pupil sample points of the most recent psf_lr call (x2, y2, xc, yc), device tensors
        self.last_pupil_points = (x2, y2, xc, yc)
        need_r = want_r and not _default_r_zero
        L, R = self._psf_buffers(out, N, ks, need_r)
        dpp = None if (dp is None or _default_r_zero) else _lib.DpParams(*[float(v) for v in dp])
        dp_ref = C.byref(dpp) if dpp is not None else None
        handle = self.dev_lens(wvln)
        flags = (_lib.PSF_NORMALIZE if normalize else 0) | self._psf_flags() \
            | (_lib.PSF_INTERLEAVED if torch.is_tensor(out) else 0)
        wkey = round(float(wvln if wvln < 10 else wvln * 1e-3), 6)
        K = len(self.surfaces)
        if center:
            # chief-ray pass (always through the green lens, optics.py:900) and primary pass in
            # one C call -- one kernel launch when a workgroup owns a point -- and ONE
            # verification round for both trip tables
            handle_c = self.dev_lens(DEFAULT_WAVE)
            MS = _lib.MAX_SURFACES
            # one control block: [primary masks | chief-ray masks | any-valid flag] -> one readback
            reference = self.trip_policy == "reference"
            verified = (reference and self.mask_reduce is None and N > 0
                        and self._spp_slices(N, spp) > 1)
            ctl = None if verified else self._zeroed_control_block(2 * MS + 1)

            def enqueue2(tp, tc):
                if N == 0:
                    return
                masks, anyv = ctl[:2 * MS].view(2, MS), ctl[2 * MS:]
                with self._timed("psf_lr_centered"):
                    _lib.check(_lib.lib().sdirt_psf_lr_centered(
                        handle, handle_c, dptr(po), N, dptr(x2), dptr(y2), spp, dptr(xc), dptr(yc),
                        xc.shape[0], float(pupilz), float(self.d_sensor), float(self.pixel_size), ks,
                        dp_ref, (C.c_int32 * K)(*[int(t) for t in tp]),
                        (C.c_int32 * K)(*[int(t) for t in tc]), flags, dptr(cen), dptr(anyv),
                        dptr(L), dptr(R), dptr(masks[0]) if reference else None,
                        dptr(masks[1]) if reference else None, stream_ptr(self.device)))

            def squeeze(L_, R_):
                if R_ is None and want_r:
                    R_ = torch.zeros_like(L_)
                if single_point:
                    L_ = L_.squeeze(0)
                    R_ = R_.squeeze(0) if R_ is not None else None
                return L_, R_

            if reference:
                keys = [("psf", wkey, self.precision), ("center", self.precision)]
                keep = (po, x2, y2, xc, yc, cen)              # alive until the kernel has run

                def enqueue_round(tables, again=True, reduce=True):
                    nonlocal ctl
                    if again:                                  # a re-launch: a fresh zeroed block
                        ctl = self._zeroed_control_block(2 * MS + 1)
                    enqueue2(tables[0], tables[1])
                    if reduce and self.mask_reduce is not None:
                        # masks AND the any-valid flag, OR-ed over ranks: every rank verifies the
                        # same block, so all of them re-launch -- or raise -- together
                        ctl.copy_(self.mask_reduce(ctl))

                def read_masks(host):
                    m = host[:2 * MS].reshape(2, MS)[:, :K].astype(np.int64) & 0xFFFFFFFF
                    return [m[0], m[1]]

                def launch(tables):
                    enqueue_round(tables)
                    host = ctl.cpu().numpy()
                    launch.any_valid = int(host[2 * MS])
                    return read_masks(host)

                def readback(block, n_words, reduce=None):
                    """Asynchronous copy of the first n_words of a control block into page-locked
                    memory on the read-back stream (on the caller's it would sit between this call's
                    kernels and the next call's); -> (host tensor, event).  reduce: the mask reduction
                    over ranks (a collective), issued on the read-back stream as well -- the caller's
                    stream goes straight on to the next call's kernel instead of waiting for it."""
                    main = torch.cuda.current_stream(self.device)
                    rb = self._side_stream("_readback_stream")
                    rb.wait_stream(main)
                    host = torch.empty(n_words, dtype=block.dtype, pin_memory=True)
                    with torch.cuda.stream(rb):
                        src = block if reduce is None else reduce(block)
                        host.copy_(src[:n_words], non_blocking=True)
                    block.record_stream(rb)
                    done = torch.cuda.Event()
                    done.record(rb)
                    return host, done

                if verified:
                    # few points, many samples (the PSFNet fitting loop): several workgroups per point.
                    # Speculate, verify ON THE DEVICE, re-render once behind it -- no host round trip
                    # when the bet on the trip tables was wrong (sdirt_psf_lr_verified).
                    W_ = _lib.CTL_WORDS
                    words = _lib.lib().sdirt_psf_verified_scratch_bytes(N, xc.shape[0]) // 4
                    scratch = self._zeroed_control_block(int(words))
                    tables = [self.trips.initial(k, self._curved()) for k in keys]
                    with self._timed("psf_lr_verified"):
                        _lib.check(_lib.lib().sdirt_psf_lr_verified(
                            handle, handle_c, dptr(po), N, dptr(x2), dptr(y2), spp, dptr(xc), dptr(yc),
                            xc.shape[0], float(pupilz), float(self.d_sensor), float(self.pixel_size), ks, dp_ref,
                            (C.c_int32 * K)(*[int(t) for t in tables[0]]),
                            (C.c_int32 * K)(*[int(t) for t in tables[1]]), flags, dptr(cen), dptr(L), dptr(R),
                            dptr(scratch), stream_ptr(self.device)))

                    def settle(h):
                        h = h.astype(np.int64) & 0xFFFFFFFF
                        launch.any_valid = int(h[_lib.CTL_ANY_VALID])
                        rounds = [(tables, [h[_lib.CTL_MASKS:_lib.CTL_MASKS + K],
                                            h[_lib.CTL_MASKS + 64:_lib.CTL_MASKS + 64 + K]])]
                        if h[_lib.CTL_STATUS]:
                            unpack = lambda w: np.array([((int(w[k >> 2]) >> ((k & 3) * 8)) & 0xFF) for k in range(K)],
                                                        np.int32).astype(np.int8).astype(np.int32)
                            t2 = [unpack(h[_lib.CTL_TRIPS2:_lib.CTL_TRIPS2 + 16]),
                                  unpack(h[_lib.CTL_TRIPS2 + 16:_lib.CTL_TRIPS2 + 32])]
                            rounds.append((t2, [h[_lib.CTL_MASKS + 128:_lib.CTL_MASKS + 128 + K],
                                                h[_lib.CTL_MASKS + 192:_lib.CTL_MASKS + 192 + K]]))
                        self.trips.run_many(keys, self._curved(), list(range(K)), launch, done=rounds)
                        assert launch.any_valid == 1, "No sampled rays is valid."   # optics.py:902
                        return squeeze(L, R) if keep else None

                    if defer:
                        host, done = readback(scratch, W_)

                        def finish_v():
                            done.synchronize()
                            return settle(host.numpy())
                        return PendingPSF(finish_v)
                    # synchronous form: into a page-locked buffer kept on the lens, then wait for the stream
                    hbuf = self.__dict__.get("_ctl_host")
                    if hbuf is None:
                        hbuf = self.__dict__["_ctl_host"] = torch.empty(W_, dtype=torch.int32, pin_memory=True)
                    hbuf.copy_(scratch[:W_], non_blocking=True)
                    torch.cuda.current_stream(self.device).synchronize()
                    return settle(hbuf.numpy().copy())

                if defer:
                    tables = [self.trips.initial(k, self._curved()) for k in keys]
                    enqueue_round(tables, again=False, reduce=False)
                    host, done = readback(ctl, 2 * MS + 1, reduce=self.mask_reduce)

                    def finish():
                        done.synchronize()
                        h = host.numpy()
                        launch.any_valid = int(h[2 * MS])
                        self.trips.run_many(keys, self._curved(), list(range(K)), launch,
                                            first=(tables, read_masks(h)))
                        assert launch.any_valid == 1, "No sampled rays is valid."   # optics.py:902
                        return squeeze(L, R) if keep else None
                    return PendingPSF(finish)
                first_round = [True]

                def launch_sync(tables):
                    enqueue_round(tables, again=not first_round[0])
                    first_round[0] = False
                    host = ctl.cpu().numpy()
                    launch.any_valid = int(host[2 * MS])
                    return read_masks(host)
                self.trips.run_many(keys, self._curved(), list(range(K)), launch_sync)
                assert launch.any_valid == 1, "No sampled rays is valid."   # optics.py:902
            else:
                full = self._fixed_trips()
                enqueue2(full, full)
            if defer:
                return PendingPSF(lambda: squeeze(L, R))
        else:
            if defer:
                raise ValueError("defer=True needs center=True")
            def enqueue(trips, mask_ptr):
                with self._timed("psf_lr"):
                    _lib.check(_lib.lib().sdirt_psf_lr(
                        handle, dptr(po), N, dptr(x2), dptr(y2), spp, float(pupilz),
                        float(self.d_sensor), float(self.pixel_size), ks, dptr(cen), dp_ref, trips,
                        flags, dptr(L), dptr(R), mask_ptr, stream_ptr(self.device)))
            self._run_with_trips(("psf", wkey, self.precision), range(K), enqueue)
        if R is None and want_r:
            R = torch.zeros_like(L)
        if single_point:
            L = L.squeeze(0)
            R = R.squeeze(0) if R is not None else None
        return L, R

    def _psf_lr_staged(self, points, po, N, ks, wvln, spp, center, dp, normalize, want_r, default_r_zero,
                       pupil_xy, center_pupil_xy, out, center_out, single_point):
        """psf_lr for SDIRT_MAX_KS < ks <= SDIRT_MAX_KS_STAGED: the reference's own sequence of optics.py:962-987
        as five library calls -- sample_from_points, psf_center, trace2sensor, forward_integral (adds into the
        grids in HBM instead of LDS), normalise -- on [spp, N] rays held in HBM (28 bytes per ray).  Same rays,
        centres and trip tables as the fused kernel; the plots that ask for such grids (draw_mtf) trace a
        handful of points."""
        if not 2 <= ks <= _lib.MAX_KS_STAGED:
            raise _lib.SdirtError(f"ks={ks} outside [2,{_lib.MAX_KS_STAGED}]")
        pupilz, pupilr = self.entrance_pupil()
        if pupil_xy is None:
            x2, y2 = self._pupil_samples(spp, pupilr)                       # optics.py:963 (first two draws)
        else:
            x2, y2 = [torch.as_tensor(v).to(self.device, torch.float32).contiguous() for v in pupil_xy]
            spp = x2.shape[0]
        ray = Ray.empty((spp, N), wvln, self.device)
        _lib.check(_lib.lib().sdirt_sample_rays(dptr(po), N, dptr(x2), dptr(y2), spp, float(pupilz), ray.c_rays(),
                                                stream_ptr(self.device)))
        cen = self._centre_buffer(center_out, N)
        xc = yc = None
        if center:
            pupilz_c, pupilr_c = self.entrance_pupil(shrink_pupil=True)
            if center_pupil_xy is None:
                xc, yc = self._pupil_samples(GEO_SPP, pupilr_c)             # optics.py:969 (draws three and four)
            else:
                xc, yc = [torch.as_tensor(v).to(self.device, torch.float32).contiguous() for v in center_pupil_xy]
            self._chief_center(po, xc, yc, pupilz_c, cen)
        else:
            pts = points.to(self.device, torch.float32)
            cen[:, 0] = pts[:, 0] * (self.sensor_size[1] / 2)               # optics.py:973-975
            cen[:, 1] = pts[:, 1] * (self.sensor_size[0] / 2)
        self.last_pupil_points = (x2, y2, xc, yc)
        self.trace(ray, forward=True, _to_sensor=self.d_sensor)        # trace2sensor, one pass
        need_r = want_r and not default_r_zero
        L, R = self._psf_buffers(out, N, ks, need_r)
        dpp = None if (dp is None or default_r_zero) else _lib.DpParams(*[float(v) for v in dp])
        h, st = _lib.lib(), stream_ptr(self.device)
        with self._timed("forward_integral"):
            # normalised on the way out of the tiles (optics.py:983-987): no second pass over the grids
            _lib.check(h.sdirt_forward_integral(ray.c_rays(), spp, N, float(self.pixel_size), int(ks), dptr(cen),
                                                C.byref(dpp) if dpp is not None else None,
                                                self._math_flags() | (_lib.PSF_NORMALIZE if normalize else 0),
                                                dptr(L), dptr(R), st))
        if R is None and want_r:
            R = torch.zeros_like(L)
        if single_point:
            L = L.squeeze(0)
            R = R.squeeze(0) if R is not None else None
        return L, R

    def _ctl_host_slot(self):
        """A page-locked control-block mirror for one call in flight (ring of eight; a slot whose call has not been waited
        for yet is settled first -- its numbers are read before anybody may overwrite them)."""
        ring = self.__dict__.setdefault("_ctl_host_ring", {"slots": [], "next": 0})
        if len(ring["slots"]) < 8:
            ring["slots"].append([torch.empty(_lib.CTL_WORDS, dtype=torch.int32, pin_memory=True), None])
            return ring["slots"][-1]
        slot = ring["slots"][ring["next"]]
        ring["next"] = (ring["next"] + 1) % 8
        if slot[1] is not None:
            slot[1].wait()
        return slot

    def _psf_call_one(self, points, po, N, ks, wvln, spp, dp, normalize, want_r, default_r_zero, out, center_out,
                      single_point, defer=False):
        """psf_lr(center=True) for sdirt_psf_spp_slices(N, spp) > 1 through sdirt_psf_call (defer: the check of what the
        device did moves into the returned PendingPSF, which waits for the call's own event instead of the stream):
        the host draws the 2 spp + 2 x 2048 uniforms (one torch.rand, the reference's order) into page-locked
        memory; ONE library call enqueues their upload, the two pupil mappings, both rounds of the
        device-verified render and the copy of the control block back; the host waits for the stream and
        checks what the device did (TripPlanner.run_many(done=...)).  Same results as the general path."""
        h, K, MS = _lib.lib(), len(self.surfaces), _lib.MAX_SURFACES
        Sc = GEO_SPP
        pupilz, pupilr = self.entrance_pupil()
        pupilr_c = self.entrance_pupil(shrink_pupil=True)[1]
        need_r = want_r and not default_r_zero
        cen = self._centre_buffer(center_out, N)
        L, R = self._psf_buffers(out, N, ks, need_r)
        dpp = None if (dp is None or default_r_zero) else _lib.DpParams(*[float(v) for v in dp])
        dp_ref = C.byref(dpp) if dpp is not None else None
        handle, handle_c = self.dev_lens(wvln), self.dev_lens(DEFAULT_WAVE)
        flags = (_lib.PSF_NORMALIZE if normalize else 0) | self._psf_flags() \
            | (_lib.PSF_INTERLEAVED if torch.is_tensor(out) else 0)
        wkey = round(float(wvln if wvln < 10 else wvln * 1e-3), 6)
        keys = [("psf", wkey, self.precision), ("center", self.precision)]
        curved = self._curved()
        tables = [self.trips.initial(k, curved) for k in keys]
        # a long streak of right bets (a caller rendering the same batch again and again): the correction
        # round is not even enqueued -- if the device's check fails after all, the host launches it (below)
        streak = self.__dict__.get("_right_streak", 0)
        if streak >= 32:
            flags |= _lib.PSF_ONE_ROUND
        n = 2 * (spp + Sc)
        words = int(h.sdirt_psf_call_scratch_bytes(N, spp, Sc) // 4)
        scratch = self._zeroed_control_block(words)
        slot = self._ctl_host_slot()
        hbuf = slot[0]
        stage, uploaded = self._staging(n, rows=1)
        st = stream_ptr(self.device)
        _hostrng.rand_into(stage[0])                         # the reference's four draws, in one (see _pupil_samples_pair)
        _lib.check(h.sdirt_psf_call(
            handle, handle_c, dptr(po), N, C.c_void_p(stage.data_ptr()), spp, Sc, float(pupilr), float(pupilr_c),
            float(pupilz), float(self.d_sensor), float(self.pixel_size), ks, dp_ref,
            (C.c_int32 * K)(*[int(t) for t in tables[0]]), (C.c_int32 * K)(*[int(t) for t in tables[1]]), flags,
            dptr(cen), dptr(L), dptr(R), dptr(scratch), C.c_void_p(hbuf.data_ptr()), st))
        stream = torch.cuda.current_stream(self.device)
        uploaded.record(stream)
        done = torch.cuda.Event()
        done.record(stream)
        keep = (po, scratch, stage, cen)                      # alive until the call has run

        def finish():
            done.synchronize()
            hh = hbuf.numpy().view(np.uint32).copy()
            slot[1] = None
            # the pupil points the device mapped: behind the control block and the partial sums in `scratch`
            base = words - 2 * n
            xy = keep[1][base + n:].view(torch.float32)
            x2, y2, xc, yc = xy[:spp], xy[spp:2 * spp], xy[2 * spp:2 * spp + Sc], xy[2 * spp + Sc:]
            self.last_pupil_points = (x2, y2, xc, yc)
            state = {"any": int(hh[_lib.CTL_ANY_VALID])}
            self.__dict__["_right_streak"] = self.__dict__.get("_right_streak", 0) + 1 if hh[_lib.CTL_STATUS] == 0 else 0
            rounds = [(tables, [hh[_lib.CTL_MASKS:_lib.CTL_MASKS + K], hh[_lib.CTL_MASKS + 64:_lib.CTL_MASKS + 64 + K]])]
            if hh[_lib.CTL_STATUS] and not (flags & _lib.PSF_ONE_ROUND):
                unpack = lambda w: np.array([((int(w[k >> 2]) >> ((k & 3) * 8)) & 0xFF) for k in range(K)],
                                            np.int32).astype(np.int8).astype(np.int32)
                rounds.append(([unpack(hh[_lib.CTL_TRIPS2:_lib.CTL_TRIPS2 + 16]),
                                unpack(hh[_lib.CTL_TRIPS2 + 16:_lib.CTL_TRIPS2 + 32])],
                               [hh[_lib.CTL_MASKS + 128:_lib.CTL_MASKS + 128 + K],
                                hh[_lib.CTL_MASKS + 192:_lib.CTL_MASKS + 192 + K]]))

            def launch(tabs):                                    # a further, host-driven round (rare)
                ctl = self._zeroed_control_block(2 * MS + 1)
                masks, anyv = ctl[:2 * MS].view(2, MS), ctl[2 * MS:]
                _lib.check(h.sdirt_psf_lr_centered(
                    handle, handle_c, dptr(po), N, dptr(x2), dptr(y2), spp, dptr(xc), dptr(yc), Sc, float(pupilz),
                    float(self.d_sensor), float(self.pixel_size), ks, dp_ref, (C.c_int32 * K)(*[int(t) for t in tabs[0]]),
                    (C.c_int32 * K)(*[int(t) for t in tabs[1]]), flags, dptr(cen), dptr(anyv), dptr(L), dptr(R),
                    dptr(masks[0]), dptr(masks[1]), stream_ptr(self.device)))
                host = ctl.cpu().numpy()
                state["any"] = int(host[2 * MS])
                m = host[:2 * MS].reshape(2, MS)[:, :K].astype(np.int64) & 0xFFFFFFFF
                return [m[0], m[1]]
            self.trips.run_many(keys, curved, list(range(K)), launch, done=rounds)
            assert state["any"] == 1, "No sampled rays is valid."   # optics.py:902
            L_, R_ = L, R
            if R_ is None and want_r:
                R_ = torch.zeros_like(L_)
            if single_point:
                L_ = L_.squeeze(0)
                R_ = R_.squeeze(0) if R_ is not None else None
            return L_, R_

        pending = PendingPSF(finish)
        slot[1] = pending
        return pending if defer else pending.wait()

    def psf_rgb(self, points, ks=31, spp=GEO_SPP, center=True, param_list=None, pupil_xy=None,
                center_pupil_xy=None):
        """optics.py:999-1015: [N,3,ks,ks] (or [3,ks,ks]) -- the three wavelengths of WAVE_RGB, each an
        independent psf_diff (fresh pupil draws, in the reference's order; every chief-ray centre
        through the green lens).  The three calls are ONE kernel launch whatever the number of points
        (center=True: sdirt_psf_rgb_centered; center=False: sdirt_psf_rgb -- lens table, pupil sets, trip
        tables and mask rows indexed by blockIdx.y) and one control-block readback.
        pupil_xy [2, 3, spp] / center_pupil_xy [2, 3, 2048]: explicit pupil sample points per
        wavelength instead of random draws (ray-level parity hand-off)."""
        if not torch.is_tensor(points):
            points = torch.tensor(points)
        n_points = points.shape[0] if points.dim() == 2 else 1
        fused = n_points > 0 and self.device.type == "cuda" and self.mask_reduce is None and ks <= _lib.MAX_KS
        if fused and center:
            return self._psf_rgb_fused(points, ks, spp, param_list, pupil_xy, center_pupil_xy)
        if fused:
            if center_pupil_xy is not None:
                raise ValueError("center=False runs no chief-ray pass")
            return self._psf_rgb_uncentred(points, ks, spp, param_list, pupil_xy)
        if pupil_xy is not None or center_pupil_xy is not None:
            # not one launch (grids above SDIRT_MAX_KS, or a rank of a sharded run): wavelength by wavelength through
            # psf_lr, which takes the explicit points of that wavelength on every path
            if center_pupil_xy is not None and not center:
                raise ValueError("center=False runs no chief-ray pass")
            dp, direct = (None, "l") if param_list is None else (tuple(param_list[:4]), param_list[4])
            psfs = []
            for i, w in enumerate(WAVE_RGB):
                lr = self.psf_lr(points, ks=ks, wvln=w, spp=spp, center=center, dp=dp,
                                 want_r=(param_list is not None and direct != "l"), _default_r_zero=(param_list is None),
                                 pupil_xy=None if pupil_xy is None else (pupil_xy[0][i], pupil_xy[1][i]),
                                 center_pupil_xy=None if center_pupil_xy is None else (center_pupil_xy[0][i], center_pupil_xy[1][i]))
                psfs.append(lr[0] if direct == "l" else lr[1])
            return torch.stack(psfs, dim=-3)
        psfs = [self.psf_diff(points=points, wvln=w, ks=ks, spp=spp, center=center,
                              param_list=param_list) for w in WAVE_RGB]
        return torch.stack(psfs, dim=-3)

    @torch.no_grad()
    def _psf_rgb_fused(self, points, ks, spp, param_list, pupil_xy=None, center_pupil_xy=None):
        single_point = points.dim() == 1
        pts = points.reshape(-1, 3)
        N, W, K, MS = pts.shape[0], len(WAVE_RGB), len(self.surfaces), _lib.MAX_SURFACES
        direct, dp_ref, dpp = "l", None, None
        if param_list is not None:
            h, f, w_, r, direct = param_list
            dpp = _lib.DpParams(float(h), float(f), float(w_), float(r))
            dp_ref = C.byref(dpp)
        want_r = direct != "l"
        po = self._points_to_object(pts)
        pupilz, pupilr = self.entrance_pupil()
        _, pupilr_c = self.entrance_pupil(shrink_pupil=True)
        # the reference's draw order: per wavelength, primary samples then chief-ray samples
        prim, cent = [], []
        for _ in WAVE_RGB:
            if pupil_xy is None:
                prim.append(torch.stack(self._pupil_samples(spp, pupilr)))
            if center_pupil_xy is None:
                cent.append(torch.stack(self._pupil_samples(GEO_SPP, pupilr_c)))
        as_dev = lambda v: torch.as_tensor(v).to(self.device, torch.float32).contiguous()
        prim = torch.stack(prim, 1).contiguous() if pupil_xy is None else as_dev(pupil_xy)              # [2, W, S]
        cent = torch.stack(cent, 1).contiguous() if center_pupil_xy is None else as_dev(center_pupil_xy)
        spp = prim.shape[2]
        handles = (C.c_void_p * W)(*[self.dev_lens(w).value for w in WAVE_RGB])
        handle_c = self.dev_lens(DEFAULT_WAVE)                # optics.py:900: always green
        cen = torch.empty((W, N, 2), dtype=torch.float32, device=self.device)
        L = torch.empty((N, W, ks, ks), dtype=torch.float32, device=self.device)
        R = torch.empty_like(L) if want_r else None
        flags = _lib.PSF_NORMALIZE | self._psf_flags()
        # one control block: [primary masks W x MS | chief-ray masks W x MS | any-valid W]
        ctl = torch.zeros(2 * W * MS + W, dtype=torch.int32, device=self.device)
        masks = ctl[:2 * W * MS].view(2, W, MS)
        anyv = ctl[2 * W * MS:]
        reference = self.trip_policy == "reference"

        def enqueue(tables):
            tp = np.concatenate([np.asarray(t, np.int32) for t in tables[:W]])
            tc = np.concatenate([np.asarray(t, np.int32) for t in tables[W:]])
            with self._timed("psf_rgb_centered"):
                _lib.check(_lib.lib().sdirt_psf_rgb_centered(
                    handles, W, handle_c, dptr(po), N, dptr(prim[0]), dptr(prim[1]), spp,
                    dptr(cent[0]), dptr(cent[1]), GEO_SPP, float(pupilz), float(self.d_sensor),
                    float(self.pixel_size), ks, dp_ref, (C.c_int32 * (W * K))(*tp.tolist()),
                    (C.c_int32 * (W * K))(*tc.tolist()), flags, dptr(cen), dptr(anyv), dptr(L), dptr(R),
                    dptr(masks[0]) if reference else None, dptr(masks[1]) if reference else None,
                    stream_ptr(self.device)))

        if reference:
            keys = [("psf", round(float(w), 6), self.precision) for w in WAVE_RGB] + \
                   [("center", self.precision)] * W

            def launch(tables):
                ctl.zero_()
                enqueue(tables)
                host = ctl.cpu().numpy()
                launch.any_valid = host[2 * W * MS:]
                m = host[:2 * W * MS].reshape(2 * W, MS)[:, :K].astype(np.int64) & 0xFFFFFFFF
                return list(m)
            self.trips.run_many(keys, self._curved(), list(range(K)), launch)
            assert bool(np.all(launch.any_valid == 1)), "No sampled rays is valid."   # optics.py:902
        else:
            full = self._fixed_trips()
            enqueue([full] * (2 * W))
        out = R if want_r else L
        return out.squeeze(0) if single_point else out

    @torch.no_grad()
    def _psf_rgb_uncentred(self, points, ks, spp, param_list, pupil_xy=None):
        """psf_rgb(center=False) as one launch (sdirt_psf_rgb): three wavelength slots, the PSFs centred on
        the pinhole image points (optics.py:972-976), one fresh primary sample set per wavelength (two
        random vectors each, optics.py:963), one control-block readback for the three trip checks."""
        single_point = points.dim() == 1
        pts = points.reshape(-1, 3)
        N, W, K, MS = pts.shape[0], len(WAVE_RGB), len(self.surfaces), _lib.MAX_SURFACES
        direct, dp_ref, dpp = "l", None, None
        if param_list is not None:
            h, f, w_, r, direct = param_list
            dpp = _lib.DpParams(float(h), float(f), float(w_), float(r))
            dp_ref = C.byref(dpp)
        want_r = direct != "l"
        po = self._points_to_object(pts)
        pupilz, pupilr = self.entrance_pupil()
        if pupil_xy is None:
            prim = torch.stack([torch.stack(self._pupil_samples(spp, pupilr)) for _ in WAVE_RGB], 1).contiguous()
        else:
            prim = torch.as_tensor(pupil_xy).to(self.device, torch.float32).contiguous()              # [2, W, S]
        spp = prim.shape[2]
        ptd = pts.to(self.device, torch.float32)
        one = torch.stack((ptd[:, 0] * (self.sensor_size[1] / 2), ptd[:, 1] * (self.sensor_size[0] / 2)), -1)
        cen = one.unsqueeze(0).expand(W, N, 2).contiguous()                                          # the same for every colour
        handles = (C.c_void_p * W)(*[self.dev_lens(w).value for w in WAVE_RGB])
        L = torch.empty((N, W, ks, ks), dtype=torch.float32, device=self.device)
        R = torch.empty_like(L) if want_r else None
        flags = _lib.PSF_NORMALIZE | self._psf_flags()
        masks = torch.zeros((W, MS), dtype=torch.int32, device=self.device)
        reference = self.trip_policy == "reference"

        def enqueue(tables):
            tp = np.concatenate([np.asarray(t, np.int32) for t in tables])
            with self._timed("psf_rgb"):
                _lib.check(_lib.lib().sdirt_psf_rgb(
                    handles, W, dptr(po), N, dptr(prim[0]), dptr(prim[1]), spp, float(pupilz), float(self.d_sensor),
                    float(self.pixel_size), ks, dptr(cen), dp_ref, (C.c_int32 * (W * K))(*tp.tolist()), flags,
                    dptr(L), dptr(R), dptr(masks) if reference else None, stream_ptr(self.device)))

        if reference:
            keys = [("psf", round(float(w), 6), self.precision) for w in WAVE_RGB]

            def launch(tables):
                masks.zero_()
                enqueue(tables)
                m = masks.cpu().numpy()[:, :K].astype(np.int64) & 0xFFFFFFFF
                return list(m)
            self.trips.run_many(keys, self._curved(), list(range(K)), launch)
        else:
            enqueue([self._fixed_trips()] * W)
        out = R if want_r else L
        return out.squeeze(0) if single_point else out

    def psf_map(self, depth=DEPTH, grid=7, ks=51, spp=GEO_SPP, center=True):
        """optics.py:1018-1041: the RGB PSFs of a grid x grid field of point sources at one depth,
        tiled into one [3, grid*ks, grid*ks] image (what torchvision's make_grid(nrow=grid,
        padding=0) assembles: PSF i sits in tile row i // grid, tile column i % grid)."""
        field = self.point_source_grid(depth=depth, grid=grid).reshape(grid * grid, 3)
        tiles = self.psf_rgb(points=field, ks=ks, center=center, spp=spp)        # [grid^2, 3, ks, ks]
        return tiles.view(grid, grid, 3, ks, ks).permute(2, 0, 3, 1, 4).reshape(3, grid * ks, grid * ks)

    def psf2mtf(self, psf, diag=False):
        """optics.py:1043-1080: the modulation transfer along the two axes of a PSF kernel -- |FFT| of its
        centre row (sagittal) and centre column (tangential), each normalised to its maximum, at the
        positive frequencies [cycles/mm] of a pixel_size sampling.  -> (freq, tangential, sagittal), numpy."""
        k = psf.detach().cpu().numpy() if torch.is_tensor(psf) else np.asarray(psf)
        row, col = k[k.shape[0] // 2, :], k[:, k.shape[1] // 2]
        sagittal, tangential = np.abs(np.fft.fft(row)), np.abs(np.fft.fft(col))
        sagittal, tangential = sagittal / sagittal.max(), tangential / tangential.max()
        freq = np.fft.fftfreq(k.shape[0], self.pixel_size)
        keep = freq > 0
        return freq[keep], tangential[keep], sagittal[keep]

    # the reference's plotting callers of the path (sdirt_amd/plots.py); each also returns what it drew
    def draw_psf_map(self, grid=9, depth=DEPTH, ks=51, log_scale=False, quater=False, save_name=None):
        """optics.py:1884-1931."""
        from . import plots
        return plots.draw_psf_map(self, grid, depth, ks, log_scale, quater, save_name)

    def draw_psf_radial(self, M=3, depth=DEPTH, ks=51, log_scale=False, save_name="./psf_radial.png"):
        """optics.py:1934-1956."""
        from . import plots
        return plots.draw_psf_radial(self, M, depth, ks, log_scale, save_name)

    def draw_mtf(self, relative_fov=[0.0, 0.7, 1.0], save_name="./mtf.png", wvlns=DEFAULT_WAVE, depth=DEPTH):
        """optics.py:2041-2067 (psf_diff at ks 256: the staged chain, _psf_lr_staged)."""
        from . import plots
        return plots.draw_mtf(self, relative_fov, save_name, wvlns, depth)

    # the lens report of the reference's fitting script and the samplers / measures under it (sdirt_amd/analysis.py)
    def analysis(self, save_name="./test", ks=None, render=False, multi_plot=False, plot_invalid=True,
                 zmx_format=False, depth=DEPTH, render_unwarp=False, lens_title=None):
        """optics.py:1663-1684."""
        from . import analysis as an
        return an.analysis(self, save_name, ks, render, multi_plot, plot_invalid, zmx_format, depth, render_unwarp,
                           lens_title)

    # ------------------------------------------------------- geometrical optics
    def calc_scale_pinhole(self, depth):
        """optics.py:1302-1306."""
        return -depth * np.tan(self.hfov) / self.r_last

    def calc_efl(self):
        """optics.py:1109-1114."""
        return self.r_last / np.tan(self.hfov)

    @torch.no_grad()
    def calc_fov(self):
        """optics.py:1203-1233: half diagonal field of view = atan of the validity-weighted mean slope
        dx/dz with which a fan of 100 rays leaves the lens towards the object, the fan starting at the
        sensor corner (r_last, 0, d_sensor) and aimed at points across the shrunk exit pupil."""
        n_fan = 100
        z_pupil, r_pupil = self.exit_pupil(shrink_pupil=True)
        corner = torch.tensor([float(self.r_last), 0.0, float(self.d_sensor)])
        aim = torch.zeros(n_fan, 3)
        aim[:, 0] = torch.linspace(-r_pupil, r_pupil, n_fan)
        aim[:, 2] = z_pupil
        fan = Ray(corner.expand(n_fan, 3), aim - corner, device=self.device)
        self.trace(fan, forward=False)
        dx, dz, weight = (fan.soa[row, :n_fan].cpu() for row in (3, 5, 6))
        half_fov = torch.atan((dx / dz * weight).sum() / weight.sum())
        if torch.isnan(half_fov):
            print("computed fov is NaN, use 0.5 rad instead.")
            return 0.5
        return half_fov.item()

    @torch.no_grad()
    def refocus(self, depth=DEPTH):
        """optics.py:1170-1196: put the sensor where green rays from the on-axis point at `depth`
        come closest to the axis.  GEO_SPP rays start on the first surface's aperture disc
        (surfaces.py:189-199; two vectors of uniforms from the CPU generator); each traced ray
        o + t d is nearest the axis at t = -(d.o)_xy / |d_xy|^2, i.e. at z = o_z - d_z (d.o)_xy /
        |d_xy|^2; the new sensor position is the mean of those z over the live rays (z > 0)."""
        front = self.surfaces[0]
        x, y = self._pupil_samples(GEO_SPP, front.r)
        start = torch.stack((x, y, torch.full_like(x, float(front.d))), dim=1)
        source = torch.tensor([0.0, 0.0, float(depth)], device=self.device)
        bundle = Ray(start, start - source, wvln=DEFAULT_WAVE, device=self.device)
        self.trace(bundle, forward=True)
        ox, oy, oz, dx, dy, dz, live = (row.cpu().numpy() for row in bundle.soa[:7, :x.shape[0]])
        with np.errstate(divide="ignore", invalid="ignore"):
            t_axis = (dx * ox + dy * oy) / (dx ** 2 + dy ** 2) * live
            z_axis = oz - dz * t_axis
        z_axis = z_axis[(live > 0) & ~np.isnan(z_axis) & (z_axis > 0)]
        d_sensor_new = float(np.mean(z_axis))
        assert d_sensor_new > 0, "sensor position is negative."
        self.d_sensor = d_sensor_new
        self.post_computation()

    def exit_pupil(self, shrink_pupil=False):
        """optics.py:1328-1332."""
        return self.entrance_pupil(entrance=False, shrink_pupil=shrink_pupil)

    def entrance_pupil(self, M=32, entrance=True, shrink_pupil=False):
        """optics.py:1379-1396 (z, r) of the paraxial pupil; the value is computed
        once per lens and cached (the reference re-traces it on every call)."""
        if self.aper_idx is None:
            s = self.surfaces[0] if entrance else self.surfaces[-1]
            return float(s.d), s.r
        if entrance not in self._pupil_cache:
            self._pupil_cache[entrance] = self.calc_entrance_pupil_paraxial(entrance=entrance)
        z, r = self._pupil_cache[entrance]
        if shrink_pupil:
            r = r * 0.25
        return z, r

    @torch.no_grad()
    def calc_entrance_pupil_paraxial(self, entrance=True):
        """optics.py:1335-1376: 16 paraxial rays from a point 1e-3 mm off-axis on the
        stop, traced out of the lens; the pairwise intersections of the emerging
        lines (in the x-z plane) give the pupil position and magnification.  The
        reference solves the 120 2x2 systems with an fp32 lstsq; here they are
        solved in closed form in float64 from the same fp32 rays."""
        aper = self.surfaces[self.aper_idx]
        aper_z, aper_r, delta_r = float(aper.d), aper.r, 1e-3
        ray_o = torch.tensor([[delta_r, 0, aper_z]]).repeat(16, 1)
        phi = torch.linspace(-0.1, 0.1, 16) / 180.0 * torch.pi
        sgn = -1.0 if entrance else 1.0
        d = torch.stack((torch.sin(phi), torch.zeros_like(phi), sgn * torch.cos(phi)), dim=-1)
        ray = Ray(ray_o, d, device=self.device)
        if entrance:
            ray, _, _ = self.trace(ray, lens_range=range(0, self.aper_idx), forward=False)
        else:
            ray, _, _ = self.trace(ray, lens_range=range(self.aper_idx + 1, len(self.surfaces)),
                                   forward=True)
        o_, d_, ra = ray.o.cpu(), ray.d.cpu(), ray.ra.cpu()
        keep = ra != 0
        O = torch.stack([o_[keep][:, 0], o_[keep][:, 2]], dim=-1)
        D = torch.stack([d_[keep][:, 0], d_[keep][:, 2]], dim=-1)
        if self.pupil_method == "reference":
            P = intersect_lines_2d_lstsq_fp32(O, D)
        else:
            P = torch.from_numpy(intersect_lines_2d(O.numpy().astype(np.float64),
                                                    D.numpy().astype(np.float64))).float()
        if len(P) == 0:
            print("No intersection points found, use the first surface as pupil.")
            return float(self.surfaces[0].d), self.surfaces[0].r
        avg_r = torch.abs((torch.mean(P[:, 0])) / delta_r * aper_r).item()
        avg_z = torch.mean(P[:, 1]).item()
        return avg_z, avg_r


def _bind_analysis(name):
    def method(self, *args, **kwargs):
        from . import analysis as an
        return getattr(an, name)(self, *args, **kwargs)
    method.__name__ = name
    method.__doc__ = f"sdirt_amd.analysis.{name} (the arguments of the reference's Lensgroup.{name})."
    return method


for _name in ("sample_parallel_2D", "sample_point_source_2D", "sample_pupil", "sample_point_source",
              "calc_magnification3", "calc_scale_ray", "analysis_rms", "calc_eqfl", "plot_setup2D", "plot_raytraces",
              "plot_setup2D_with_trace"):
    setattr(Lensgroup, _name, _bind_analysis(_name))


def intersect_lines_2d_lstsq_fp32(origins, directions):
    """optics.py:1470-1515 as the reference evaluates it: all pairs, the 2x2 systems
    [Di, -Dj] x = Oj - Oi solved in fp32 by torch.linalg.lstsq (CPU: LAPACK gelsy),
    both evaluations of the intersection averaged.  Host-side, 120 tiny systems."""
    n = origins.shape[0]
    idx_i, idx_j = torch.combinations(torch.arange(n), r=2).unbind(1)
    Oi, Oj, Di, Dj = origins[idx_i], origins[idx_j], directions[idx_i], directions[idx_j]
    b = Oj - Oi
    A = torch.stack([Di, -Dj], dim=-1)
    x = torch.linalg.lstsq(A, b.unsqueeze(-1))[0].squeeze(-1)
    P_i = Oi + x[:, 0].unsqueeze(-1) * Di
    P_j = Oj + x[:, 1].unsqueeze(-1) * Dj
    return (P_i + P_j) / 2


def intersect_lines_2d(origins, directions):
    """optics.py:1470-1515: pairwise intersections of N 2-D lines, [N(N-1)/2, 2].
    Solves Oi + s Di = Oj + t Dj exactly (float64) and averages both evaluations."""
    n = origins.shape[0]
    ii, jj = np.triu_indices(n, k=1)
    Oi, Oj, Di, Dj = origins[ii], origins[jj], directions[ii], directions[jj]
    b = Oj - Oi
    det = Di[:, 0] * (-Dj[:, 1]) - (-Dj[:, 0]) * Di[:, 1]
    ok = det != 0
    Oi, Oj, Di, Dj, b, det = Oi[ok], Oj[ok], Di[ok], Dj[ok], b[ok], det[ok]
    s = (b[:, 0] * (-Dj[:, 1]) - (-Dj[:, 0]) * b[:, 1]) / det
    t = (Di[:, 0] * b[:, 1] - Di[:, 1] * b[:, 0]) / det
    Pi = Oi + s[:, None] * Di
    Pj = Oj + t[:, None] * Dj
    return (Pi + Pj) / 2
