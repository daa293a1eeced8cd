"""ctypes front-end of the CPU parity oracle (oracle/sdirt_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under sdirt_amd/ imports this module.
Parity status: PINNED against the reference's own outputs -- see
tests/test_oracle_golden.py and the fixtures in tests/golden/.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsdirt_oracle.so")

KIND = {"plane": 0, "sphere": 1, "asphere": 2}


class OrSurface(C.Structure):
    _fields_ = [("kind", C.c_int32), ("ai_degree", C.c_int32), ("r", C.c_float),
                ("d", C.c_float), ("c", C.c_float), ("k", C.c_float),
                ("ai", C.c_float * 8), ("r_d", C.c_double), ("n1", C.c_double),
                ("n2", C.c_double)]


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "sdirt_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "-B", "libsdirt_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        assert _lib.or_sizeof_surface() == C.sizeof(OrSurface)
        _lib.or_num_threads.restype = C.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def load_state(path_or_name):
    """Lens state fixture (tests/golden/lens_state_*.json, written from the
    reference by gen_golden.py)."""
    p = path_or_name
    if not os.path.exists(p):
        p = os.path.join(HERE, "..", "tests", "golden", f"lens_state_{path_or_name}.json")
    with open(p) as f:
        return json.load(f)


def surfaces_from_state(state, wvln):
    """-> ctypes array of OrSurface with n1/n2 at `wvln` (key repr(float))."""
    key = repr(float(wvln))
    arr = (OrSurface * len(state["surfaces"]))()
    for i, s in enumerate(state["surfaces"]):
        a = arr[i]
        a.kind = KIND[s["kind"]]
        a.ai_degree = len(s["ai"])
        a.r = s["r"]; a.r_d = s["r"]
        a.d = s["d"]; a.c = s["c"]; a.k = s["k"]
        for j, v in enumerate(s["ai"]):
            a.ai[j] = v
        a.n1 = s["n1"][key]; a.n2 = s["n2"][key]
    return arr


def points_to_object(points, state):
    pts = _f32(points).reshape(-1, 3)
    out = np.empty_like(pts)
    lib().or_points_to_object(_fp(pts), C.c_int64(len(pts)), C.c_double(np.tan(state["hfov"])),
                              C.c_double(state["r_last"]), C.c_double(state["sensor_size"][1]),
                              C.c_double(state["sensor_size"][0]), _fp(out))
    return out


def pupil_samples(u_theta, u_r2, pupil_r):
    ut, ur = _f32(u_theta), _f32(u_r2)
    x2, y2 = np.empty_like(ut), np.empty_like(ut)
    lib().or_pupil_samples(_fp(ut), _fp(ur), C.c_int64(len(ut)), C.c_double(pupil_r),
                           _fp(x2), _fp(y2))
    return x2, y2


def sample_rays(point_obj, x2, y2, pupil_z):
    po, x2, y2 = _f32(point_obj).reshape(-1, 3), _f32(x2), _f32(y2)
    S, N = len(x2), len(po)
    o = np.empty((S, N, 3), np.float32); d = np.empty((S, N, 3), np.float32)
    ra = np.empty((S, N), np.float32); ob = np.empty((S, N), np.float32)
    lib().or_sample_rays(_fp(po), C.c_int64(N), _fp(x2), _fp(y2), C.c_int64(S),
                         C.c_double(pupil_z), _fp(o), _fp(d), _fp(ra), _fp(ob))
    return o, d, ra, ob


def trace(surf, o, d, ra, obliq=None, first=0, last=None, trips=None, record=False):
    """In-place on copies; returns dict(o,d,ra,obliq,trips[,rec_o,rec_d,rec_ra])."""
    K = len(surf)
    last = K if last is None else last
    o, d, ra = _f32(o).copy(), _f32(d).copy(), _f32(ra).copy()
    ob = np.ones_like(ra) if obliq is None else _f32(obliq).copy()
    M = ra.size
    tr = np.full(K, -1, np.int32) if trips is None else np.ascontiguousarray(trips, np.int32).copy()
    nrec = last - first
    rec = [None] * 3
    if record:
        rec = [np.empty((nrec,) + o.shape, np.float32), np.empty((nrec,) + d.shape, np.float32),
               np.empty((nrec,) + ra.shape, np.float32)]
    lib().or_trace(surf, C.c_int(first), C.c_int(last), C.c_int64(M), _fp(o), _fp(d), _fp(ra),
                   _fp(ob), tr.ctypes.data_as(C.POINTER(C.c_int32)),
                   *[(_fp(r) if r is not None else None) for r in rec])
    out = dict(o=o, d=d, ra=ra, obliq=ob, trips=tr)
    if record:
        out.update(rec_o=rec[0], rec_d=rec[1], rec_ra=rec[2])
    return out


def propagate_to(z, o, d):
    o = _f32(o).copy(); d = _f32(d)
    lib().or_propagate_to(C.c_double(z), C.c_int64(o.size // 3), _fp(o), _fp(d))
    return o


def center_from_rays(o, ra):
    o, ra = _f32(o), _f32(ra)
    S, N = ra.shape
    c = np.empty((N, 2), np.float32)
    lib().or_center_from_rays.restype = C.c_int
    ok = lib().or_center_from_rays(C.c_int64(S), C.c_int64(N), _fp(o), _fp(ra), _fp(c))
    return c, bool(ok)


def _dp(dp):
    if dp is None:
        return None
    return (C.c_double * 4)(*[float(v) for v in dp[:4]])


def assign_points_to_pixels(points, ra, x_tan, ks, x_range, dp=None):
    pts, ra, xt = _f32(points), _f32(ra), _f32(x_tan)
    lg = np.empty((ks, ks), np.float32); rg = np.empty((ks, ks), np.float32)
    lib().or_assign_points_to_pixels(_fp(pts), _fp(ra), _fp(xt), C.c_int64(len(ra)),
                                     C.c_int64(1), C.c_int(ks), C.c_double(x_range[0]),
                                     C.c_double(x_range[1]), _dp(dp), _fp(lg), _fp(rg))
    return lg, rg


def forward_integral(o, d, ra, ps, ks, center, dp=None):
    o, d, ra, center = _f32(o), _f32(d), _f32(ra), _f32(center)
    S, N = ra.shape
    lg = np.empty((N, ks, ks), np.float32); rg = np.empty((N, ks, ks), np.float32)
    lib().or_forward_integral(C.c_int64(S), C.c_int64(N), _fp(o), _fp(d), _fp(ra),
                              C.c_double(ps), C.c_int(ks), _fp(center), _dp(dp), _fp(lg), _fp(rg))
    return lg, rg


def psf_normalize(psf):
    p = _f32(psf).copy()
    N, ks, _ = p.shape
    lib().or_psf_normalize(C.c_int64(N), C.c_int(ks), _fp(p))
    return p


def psf(state, points, x2, y2, xc, yc, ks, wvln=0.589, dp=None, normalize=True,
        center_wvln=0.589, return_trips=False):
    """End-to-end psf_diff on explicit pupil samples -> (L, R, centre, ok); with return_trips
    also the batch-global Newton trip tables (primary, chief-ray) the reference's loop
    (surfaces.py:547) runs on this batch."""
    surf = surfaces_from_state(state, wvln)
    surf_c = surfaces_from_state(state, center_wvln)
    po = points_to_object(points, state)
    x2, y2, xc, yc = _f32(x2), _f32(y2), _f32(xc), _f32(yc)
    N, K = len(po), len(surf)
    cen = np.empty((N, 2), np.float32)
    lg = np.empty((N, ks, ks), np.float32); rg = np.empty((N, ks, ks), np.float32)
    tp, tc = np.zeros(K, np.int32), np.zeros(K, np.int32)
    i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    lib().or_psf_trips.restype = C.c_int
    ok = lib().or_psf_trips(surf, surf_c, C.c_int(K), _fp(po), C.c_int64(N), _fp(x2), _fp(y2),
                            C.c_int64(len(x2)), _fp(xc), _fp(yc), C.c_int64(len(xc)),
                            C.c_double(state["pupil_z"]), C.c_double(state["d_sensor"]),
                            C.c_double(state["pixel_size"]), C.c_int(ks), _dp(dp),
                            C.c_int(1 if normalize else 0), _fp(cen), _fp(lg), _fp(rg),
                            i32(tp), i32(tc))
    if return_trips:
        return lg, rg, cen, bool(ok), tp, tc
    return lg, rg, cen, bool(ok)


def num_threads():
    return lib().or_num_threads()


def set_num_threads(n):
    lib().or_set_num_threads(C.c_int(n))
