"""PyTorch-CPU restatement of the dual-pixel PSF path -- the `torch` leg of bench.py's cpu_baseline.

TEST / BASELINE INFRASTRUCTURE ONLY (same rule as oracle/sdirt_oracle.c): imported by tests/ and by
the cpu_baseline leg of bench.py, never by sdirt_amd/.

What it is: the reference's EXECUTION MODEL (LinYark/Sdirt runs this path as whole-tensor fp32
PyTorch ops on the host: one op per arithmetic step over an [spp, N] ray tensor, a batch-wide
`.any()` deciding every Newton trip, `index_put_(accumulate=True)` for the splat) written from
scratch by the builder -- the reference's files never travel to the GPU box.  It follows the same
operation sequence as oracle/sdirt_oracle.c (which cites the reference line by line:
deeplens/optics.py:934-996, surfaces.py:391-830, monte_carlo.py:9-372) but is organised
differently from the reference: component-wise SoA tensors instead of [..., 3] AoS, one flat
scatter per bilinear tap for ALL points instead of a Python loop over points, explicit products
instead of `pow`.  It is therefore at least as fast as the reference on the same cores
(cross-checked against BASELINE.md's probe timings in DESIGN.md §7) and agrees with the reference
fixtures to ~1e-4 of the PSF peak (tests/test_oracle_golden.py::test_torch_port_*): it is a
TIMING baseline, the bit-level parity oracle is the C file.
"""
import math

import numpy as np
import torch

TOL_LOOSE, TOL_TIGHT, STEP_BOUND, MAXITER, EPS = 50e-6, 10e-6, 5.0, 10, 1e-9


class _Surf:
    """Per-surface constants, rounded to fp32 where torch's scalar handling rounds them."""

    def __init__(self, s, key):
        self.kind = s["kind"]
        self.r = float(s["r"])
        self.d = float(np.float32(s["d"]))
        self.c = float(np.float32(s["c"]))
        self.k = float(np.float32(s["k"]))
        self.ai = [float(np.float32(a)) for a in s["ai"]] if self.kind == "asphere" else []
        self.n1, self.n2 = float(s["n1"][key]), float(s["n2"][key])
        if self.kind != "plane":
            c2 = np.float32(self.c) * np.float32(self.c)
            self.c2 = float(c2)
            self.onepk = float(np.float32(1.0) + np.float32(self.k))
            self.lim_loose = float(np.float32(1.0) / c2 * np.float32(1.0 - EPS) / np.float32(self.onepk))
            self.r2_lim = float(np.float32(self.r * self.r))


def surfaces(state, wvln):
    key = repr(float(wvln))
    return [_Surf(s, key) for s in state["surfaces"]]


def points_to_object(points, state):
    p = torch.as_tensor(points, dtype=torch.float32).reshape(-1, 3)
    depth = p[:, 2]
    scale = (-depth) * float(np.float32(math.tan(state["hfov"]))) / float(np.float32(state["r_last"]))
    x = p[:, 0] * scale * float(state["sensor_size"][1]) / 2.0
    y = p[:, 1] * scale * float(state["sensor_size"][0]) / 2.0
    return torch.stack((x, y, depth), dim=-1)


def pupil_samples(u_theta, u_r2, pupil_r):
    theta = torch.as_tensor(u_theta, dtype=torch.float32) * 2.0 * math.pi
    r = torch.sqrt(torch.as_tensor(u_r2, dtype=torch.float32) * float(pupil_r) ** 2)
    return r * torch.cos(theta), r * torch.sin(theta)


class Rays:
    """[S, N] component tensors."""
    __slots__ = ("ox", "oy", "oz", "dx", "dy", "dz", "ra", "ob")


def _unit(x, y, z):
    n = torch.sqrt(x * x + y * y + z * z).clamp_min_(1e-12)
    return x / n, y / n, z / n


def sample(po, x2, y2, pupil_z):
    S, N = x2.shape[0], po.shape[0]
    r = Rays()
    r.ox, r.oy, r.oz = [po[:, i].unsqueeze(0).expand(S, N).contiguous() for i in range(3)]
    r.dx, r.dy, r.dz = _unit(x2.unsqueeze(1) - r.ox, y2.unsqueeze(1) - r.oy, float(pupil_z) - r.oz)
    r.ra = torch.ones((S, N))
    r.ob = torch.ones((S, N))
    return r


def _sag(s, r2):
    """(g, dg/dr2) of the even asphere at squared radius r2."""
    a = (s.onepk * r2) * s.c2
    sf = torch.sqrt(1.0 - a)
    onesf = 1.0 + sf
    g = (r2 * s.c) / onesf
    dgd = ((onesf + (a * 0.5) / sf) * s.c) / (onesf * onesf)
    if s.ai:
        pw = torch.ones_like(r2)
        for i, ai in enumerate(s.ai):
            dgd = dgd + ((i + 1) * ai) * pw
            pw = pw * r2
            g = g + ai * pw
    return g, dgd


def _newton(s, r):
    alive = r.ra > 0
    t0 = (s.d - r.oz) / r.dz
    dd = r.dx * r.dx + r.dy * r.dy
    dox = r.dx * r.ox + r.dy * r.oy
    kgt = s.k > -1.0

    def step(t, tight):
        nx, ny, nz = r.ox + r.dx * t, r.oy + r.dy * t, r.oz + r.dz * t
        rr = nx * nx + ny * ny
        inside = (rr < s.lim_loose) if kgt else (rr > 0)
        if tight:
            inside = (rr < s.r2_lim) & inside if kgt else (rr < s.r2_lim)
        r2 = torch.where(inside & alive, rr, torch.zeros_like(rr))
        g, dgd = _sag(s, r2)
        ft = (g + s.d) - nz
        dfdt = dgd * (2.0 * (dd * t + dox)) - r.dz
        return ft, t - (ft / (dfdt + EPS)).clamp_(-STEP_BOUND, STEP_BOUND)

    t, it = t0, 0
    ft = torch.full_like(t0, 1e5)
    while it < MAXITER and bool((ft.abs() > TOL_LOOSE).any()):     # batch-wide trip decision
        it += 1
        ft, t = step(t, tight=False)
    t = t0 + (t - t0)
    ft, t = step(t, tight=True)
    nx, ny = r.ox + r.dx * t, r.oy + r.dy * t
    rr = nx * nx + ny * ny
    v = (rr < s.r2_lim) & alive & (ft.abs() < TOL_TIGHT) & (t > 0)
    if kgt:
        v = v & (rr < s.lim_loose)
    return t, v, it


def _refract(s, r, forward):
    eta = s.n1 / s.n2 if forward else s.n2 / s.n1
    if s.kind == "plane":
        nx, ny, nz = torch.zeros_like(r.ox), torch.zeros_like(r.ox), -torch.ones_like(r.ox)
    elif s.kind == "sphere":
        dR = float(np.float32(s.d) + np.float32(1.0) / np.float32(s.c))
        sg = 2.0 if s.c > 0 else -2.0
        nx, ny, nz = sg * r.ox, sg * r.oy, sg * r.oz - sg * dR
    else:
        vf = (r.ra > 0).to(torch.float32)
        xv, yv = r.ox * vf, r.oy * vf
        _, ds = _sag(s, xv * xv + yv * yv)
        nx, ny, nz = (ds * 2.0) * xv, (ds * 2.0) * yv, -torch.ones_like(xv)
    nx, ny, nz = _unit(nx, ny, nz)
    if forward:
        nx, ny, nz = -nx, -ny, -nz
    eta32, eta2 = float(np.float32(eta)), float(np.float32(eta * eta))
    cosi = (r.dx * nx + r.dy * ny) + r.dz * nz
    c2i = cosi * cosi
    omc = 1.0 - c2i
    v = (c2i > 0.1) & (eta2 * omc < 1.0) & (r.ra > 0)
    vf = v.to(torch.float32)
    sr = torch.sqrt(1.0 - (eta2 * omc) * vf)
    ndx = torch.where(v, sr * nx + eta32 * (r.dx - cosi * nx), r.dx)
    ndy = torch.where(v, sr * ny + eta32 * (r.dy - cosi * ny), r.dy)
    ndz = torch.where(v, sr * nz + eta32 * (r.dz - cosi * nz), r.dz)
    r.ob = r.ob * ((ndx * r.dx + ndy * r.dy) + ndz * r.dz)
    r.dx, r.dy, r.dz = ndx, ndy, ndz
    r.ra = r.ra * vf


def trace(surfs, r):
    """All surfaces front to back; returns the Newton trip counts."""
    trips = []
    for s in surfs:
        forward = bool((r.dz * r.ra).sum() > 0)
        if s.kind == "plane":
            t = (s.d - r.oz) / r.dz
            nx, ny, nz = r.ox + t * r.dx, r.oy + t * r.dy, r.oz + t * r.dz
            v = (torch.sqrt(nx * nx + ny * ny) <= float(np.float32(s.r))) & (r.ra > 0)
            it = 0
        else:
            t, vn, it = _newton(s, r)
            nx, ny, nz = r.ox + t * r.dx, r.oy + t * r.dy, r.oz + t * r.dz
            v = ((nx * nx + ny * ny <= s.r2_lim) & (t >= 0) & (r.ra > 0)) if s.kind == "sphere" else vn
        r.ox, r.oy, r.oz = torch.where(v, nx, r.ox), torch.where(v, ny, r.oy), torch.where(v, nz, r.oz)
        r.ra = r.ra * v.to(torch.float32)
        if s.kind != "plane" or s.n1 / s.n2 != 1.0:
            _refract(s, r, forward)
        trips.append(it)
    return trips


def propagate_to(r, z):
    t = (float(z) - r.oz) / r.dz
    r.ox, r.oy, r.oz = r.ox + r.dx * t, r.oy + r.dy * t, r.oz + r.dz * t


def center(r):
    den = r.ra.sum(0) + EPS
    return torch.stack((-((r.ox * r.ra).sum(0) / den), -((r.oy * r.ra).sum(0) / den)), dim=-1)


def _seg(u):
    return u - 0.5 * torch.sin(2.0 * u)


def _dp_weights(x_tan, dp):
    """Left / right sub-pixel areas seen through the microlens, per ray (both radius branches)."""
    h, f, w, rad = [float(np.float32(v)) for v in dp]
    fmh = float(np.float32(float(dp[1]) - float(dp[0])))
    rr = float(np.float32(rad) * np.float32(rad))
    big = float(dp[3]) > 0.5
    tr = math.asin(0.5 / rad) if big else 0.0
    tl = math.pi - tr

    def areas(xr, xm, xl, lim):
        xr, xm, xl = xr.clamp(-lim, lim), xm.clamp(-lim, lim), xl.clamp(-lim, lim)
        ur, um, ul = torch.acos(xr / rad), torch.acos(xm / rad), torch.acos(xl / rad)
        sm = _seg(um)
        a_r, a_l = rr * (sm - _seg(ur)), rr * (_seg(ul) - sm)
        if big:       # corners of the square pixel clip the microlens disc
            ure, ume, ule = ur.clamp(tr, tl), um.clamp(tr, tl), ul.clamp(tr, tl)
            xre, xme, xle = torch.cos(ure) * rad, torch.cos(ume) * rad, torch.cos(ule) * rad
            sme = _seg(ume)
            a_r = a_r - ((rr * (sme - _seg(ure))) - (xre - xme))
            a_l = a_l - ((rr * (_seg(ule) - sme)) - (xme - xle))
        return a_r, a_l

    fx = f * x_tan
    lim = 0.5 if big else rad
    sr_ml, sl_ml = areas(w - ((fx - w) * h) / fmh, ((-fx) * h) / fmh, (-w) - ((fx + w) * h) / fmh, lim)
    hx = h * x_tan
    xr, xm, xl = (w - hx).clamp(-0.5, 0.5), (0.0 - hx).clamp(-0.5, 0.5), ((-w) - hx).clamp(-0.5, 0.5)
    sr_in, sl_in = areas(xr, xm, xl, lim)
    return sl_ml + ((xm - xl) - sl_in), sr_ml + ((xr - xm) - sr_in)


def forward_integral(r, ps, ks, cen, dp=None):
    """Sensor-plane rays -> raw (L, R) [N, ks, ks]; R stays zero without dp (param_list=None)."""
    S, N = r.ra.shape
    hi, lo = (ks / 2.0 - 0.5) * ps, (-ks / 2.0 + 0.5) * ps
    lim = float(np.float32(hi - 0.01 * ps))
    px, py = (-r.ox) - cen[:, 0], (-r.oy) - cen[:, 1]
    w = r.ra * (px.abs() < lim).to(torch.float32) * (py.abs() < lim).to(torch.float32)
    px, py = px * w, py * w
    sl, sr = _dp_weights((-r.dx) / r.dz, dp if dp is not None else (0.78, 1.44, 0.3, 0.5))
    pf0 = ((py - float(np.float32(hi))) / float(np.float32(lo - hi))) * float(ks - 1)
    pf1 = ((px - float(np.float32(lo))) / float(np.float32(hi - lo))) * float(ks - 1)
    fl0, fl1 = torch.floor(pf0), torch.floor(pf1)
    wb, wr = pf0 - fl0, pf1 - fl1
    r0, c0 = fl0.long(), fl1.long()
    r1, c1 = torch.floor(pf0 + 1.0).long(), torch.floor(pf1 + 1.0).long()
    base = (torch.arange(N) * (ks * ks)).unsqueeze(0)
    L = torch.zeros(N * ks * ks)
    R = torch.zeros(N * ks * ks)
    taps = ((r0, c0, (1.0 - wb) * (1.0 - wr)), (r0, c1, (1.0 - wb) * wr),
            (r1, c0, wb * (1.0 - wr)), (r0 + 1, c0 + 1, wb * wr))
    for rows, cols, wt in taps:                 # one flat scatter-add per tap, all points at once
        idx = (base + rows * ks + cols).reshape(-1)
        wra = wt * w
        L.index_put_((idx,), (wra * sl).reshape(-1), accumulate=True)
        if dp is not None:
            R.index_put_((idx,), (wra * sr).reshape(-1), accumulate=True)
    return L.reshape(N, ks, ks), R.reshape(N, ks, ks)


@torch.no_grad()
def psf(state, points, x2, y2, xc, yc, ks, wvln=0.589, dp=None, normalize=True, chunk=256,
        center_wvln=0.589):
    """End to end on explicit pupil samples, `chunk` points per pass -> (L, R, centres, trips)."""
    surf, surf_c = surfaces(state, wvln), surfaces(state, center_wvln)
    po_all = points_to_object(points, state)
    x2, y2, xc, yc = [torch.as_tensor(v, dtype=torch.float32) for v in (x2, y2, xc, yc)]
    Ls, Rs, Cs, trips = [], [], [], None
    for a in range(0, po_all.shape[0], chunk):
        po = po_all[a:a + chunk]
        rc = sample(po, xc, yc, state["pupil_z"])
        trace(surf_c, rc)
        propagate_to(rc, state["d_sensor"])
        cen = center(rc)
        r = sample(po, x2, y2, state["pupil_z"])
        trips = trace(surf, r)
        propagate_to(r, state["d_sensor"])
        L, R = forward_integral(r, state["pixel_size"], ks, cen, dp)
        if normalize:
            L = L / (L.amax(dim=(1, 2), keepdim=True) + 1e-6)
            R = R / (R.amax(dim=(1, 2), keepdim=True) + 1e-6)
        Ls.append(L); Rs.append(R); Cs.append(cen)
    return torch.cat(Ls), torch.cat(Rs), torch.cat(Cs), trips
