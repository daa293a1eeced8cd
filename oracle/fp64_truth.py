"""Float64 evaluation of the dual-pixel PSF path on the SAME fp32 inputs -- "truth" for parity.

TEST INFRASTRUCTURE ONLY (imported by tests/ only).

The reference (LinYark/Sdirt) and the HIP kernels both run the path in fp32 and differ from each
other in the last bits of individual operations (torch's CPU sqrt / sin / cos / acos are MKL VML
kernels, < 1 ulp but not correctly rounded; the kernels' are correctly rounded).  To say which of
the two is closer to what the algorithm MEANS, this module evaluates the same operation sequence --
same fp32 input rays, same fp32 lens constants, the reference's batch-global Newton trip counts
(passed in) -- with every arithmetic operation in IEEE binary64, i.e. with rounding errors 2^29 times
smaller.  tests/test_gpu_parity.py asserts that the HIP PSFs are at least as close to it as the
reference's own PSFs are.

It follows the same operation sequence as oracle/sdirt_oracle.c, which cites the reference line by
line (deeplens/surfaces.py:391-830, monte_carlo.py:9-240, optics.py:979-987).
"""
import numpy as np

TOL_TIGHT, STEP_BOUND, EPS = 10e-6, 5.0, 1e-9


def _sag(s, r2):
    a = (1.0 + s["k"]) * r2 * s["c"] ** 2
    sf = np.sqrt(1.0 - a)
    onesf = 1.0 + sf
    g = r2 * s["c"] / onesf
    dgd = (onesf + a * 0.5 / sf) * s["c"] / (onesf * onesf)
    for i, ai in enumerate(s["ai"] if s["kind"] == "asphere" else []):
        dgd = dgd + (i + 1) * ai * r2 ** i
        g = g + ai * r2 ** (i + 1)
    return g, dgd


def _unit(v):
    return v / np.maximum(np.sqrt((v * v).sum(-1, keepdims=True)), 1e-12)


def _refract(s, key, o, d, ra, forward):
    eta = s["n1"][key] / s["n2"][key] if forward else s["n2"][key] / s["n1"][key]
    if s["kind"] == "plane":
        n = np.zeros_like(o)
        n[..., 2] = -1.0
    elif s["kind"] == "sphere":
        cen = np.array([0.0, 0.0, s["d"] + 1.0 / s["c"]])
        n = 2.0 * (o - cen) if s["c"] > 0 else -2.0 * (o - cen)
    else:
        vf = (ra > 0)[..., None]
        xy = o[..., :2] * vf
        _, ds = _sag(s, (xy * xy).sum(-1))
        n = np.concatenate([2.0 * ds[..., None] * xy, -np.ones_like(o[..., :1])], -1)
    n = _unit(n)
    if forward:
        n = -n
    cosi = (d * n).sum(-1)
    c2i = cosi * cosi
    v = (c2i > 0.1) & (eta * eta * (1.0 - c2i) < 1.0) & (ra > 0)
    sr = np.sqrt(1.0 - eta * eta * (1.0 - c2i) * v)
    nd = sr[..., None] * n + eta * (d - cosi[..., None] * n)
    return np.where(v[..., None], nd, d), ra * v


def trace(state, o, d, trips, wvln=0.589):
    """o, d [..., 3] float64 (from fp32 values), forward through every surface with the given
    Newton trip counts -> (o, d, ra)."""
    key = repr(float(wvln))
    ra = np.ones(o.shape[:-1])
    for s, T in zip(state["surfaces"], trips):
        forward = (d[..., 2] * ra).sum() > 0
        if s["kind"] == "plane":
            t = (s["d"] - o[..., 2]) / d[..., 2]
            new = o + t[..., None] * d
            v = (np.sqrt((new[..., :2] ** 2).sum(-1)) <= s["r"]) & (ra > 0)
        else:
            kgt = s["k"] > -1.0
            lim_loose = (1.0 - EPS) / s["c"] ** 2 / (1.0 + s["k"]) if kgt else None
            r2_lim = s["r"] ** 2
            dd = (d[..., :2] ** 2).sum(-1)
            dox = (d[..., :2] * o[..., :2]).sum(-1)

            def step(t, tight):
                p = o + t[..., None] * d
                rr = (p[..., :2] ** 2).sum(-1)
                inside = (rr < lim_loose) if kgt else (rr > 0)
                if tight:
                    inside = (rr < r2_lim) & (inside if kgt else True)
                g, dgd = _sag(s, np.where(inside & (ra > 0), rr, 0.0))
                ft = g + s["d"] - p[..., 2]
                dfdt = dgd * 2.0 * (dd * t + dox) - d[..., 2]
                return ft, t - np.clip(ft / (dfdt + EPS), -STEP_BOUND, STEP_BOUND)

            t0 = (s["d"] - o[..., 2]) / d[..., 2]
            t = t0
            for _ in range(int(T)):
                _, t = step(t, False)
            ft, t = step(t, True)
            new = o + t[..., None] * d
            rr = (new[..., :2] ** 2).sum(-1)
            if s["kind"] == "sphere":
                v = (rr <= r2_lim) & (t >= 0) & (ra > 0)
            else:
                v = (rr < r2_lim) & (ra > 0) & (np.abs(ft) < TOL_TIGHT) & (t > 0)
                if kgt:
                    v = v & (rr < lim_loose)
        o = np.where(v[..., None], new, o)
        ra = ra * v
        if s["kind"] != "plane" or s["n1"][key] != s["n2"][key]:
            d, ra = _refract(s, key, o, d, ra, forward)
    return o, d, ra


def to_sensor(state, o, d):
    t = (state["d_sensor"] - o[..., 2]) / d[..., 2]
    return o + t[..., None] * d


def center(o_sensor, ra):
    den = ra.sum(0) + EPS
    return -np.stack(((o_sensor[..., 0] * ra).sum(0) / den, (o_sensor[..., 1] * ra).sum(0) / den), -1)


def _seg(u):
    return u - 0.5 * np.sin(2.0 * u)


def forward_integral(o_sensor, d, ra, ps, ks, cen, dp):
    """Raw (L, R) [N, ks, ks] in float64 (small-radius microlens branch)."""
    S, N = ra.shape
    h, f, w, rad = [float(v) for v in dp]
    hi, lo = (ks / 2.0 - 0.5) * ps, (-ks / 2.0 + 0.5) * ps
    lim = hi - 0.01 * ps
    px, py = -o_sensor[..., 0] - cen[:, 0], -o_sensor[..., 1] - cen[:, 1]
    wgt = ra * (np.abs(px) < lim) * (np.abs(py) < lim)
    px, py = px * wgt, py * wgt
    x_tan = -d[..., 0] / d[..., 2]

    def areas(xr, xm, xl):
        ur, um, ul = [np.arccos(np.clip(v, -rad, rad) / rad) for v in (xr, xm, xl)]
        return rad * rad * (_seg(um) - _seg(ur)), rad * rad * (_seg(ul) - _seg(um))

    fx = f * x_tan
    sr_ml, sl_ml = areas(w - (fx - w) * h / (f - h), -fx * h / (f - h), -w - (fx + w) * h / (f - h))
    hx = h * x_tan
    xr, xm, xl = [np.clip(v, -0.5, 0.5) for v in (w - hx, -hx, -w - hx)]
    sr_in, sl_in = areas(xr, xm, xl)
    sl, sr = sl_ml + (xm - xl) - sl_in, sr_ml + (xr - xm) - sr_in
    pf0 = (py - hi) / (lo - hi) * (ks - 1)
    pf1 = (px - lo) / (hi - lo) * (ks - 1)
    r0, c0 = np.floor(pf0).astype(np.int64), np.floor(pf1).astype(np.int64)
    wb, wr = pf0 - r0, pf1 - c0
    base = (np.arange(N) * ks * ks)[None, :]
    L, R = np.zeros(N * ks * ks), np.zeros(N * ks * ks)
    for dr, dc, wt in ((0, 0, (1 - wb) * (1 - wr)), (0, 1, (1 - wb) * wr), (1, 0, wb * (1 - wr)), (1, 1, wb * wr)):
        rows, cols = np.minimum(r0 + dr, ks - 1), np.minimum(c0 + dc, ks - 1)
        idx = (base + rows * ks + cols).reshape(-1)
        np.add.at(L, idx, (wt * wgt * sl).reshape(-1))
        np.add.at(R, idx, (wt * wgt * sr).reshape(-1))
    return L.reshape(N, ks, ks), R.reshape(N, ks, ks)


def normalize(psf):
    return psf / (psf.max(axis=(1, 2), keepdims=True) + 1e-6)


def psf_from_rays(state, o0, d0, trips, cen, ks, dp, wvln=0.589):
    """fp32 rays (o0, d0 [S, N, 3]) + trip table + centres -> max-normalised (L, R) in float64."""
    o, d, ra = trace(state, o0.astype(np.float64), d0.astype(np.float64), trips, wvln)
    osen = to_sensor(state, o, d)
    L, R = forward_integral(osen, d, ra, state["pixel_size"], ks, cen.astype(np.float64), dp)
    return normalize(L), normalize(R)
