#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

TEST INFRASTRUCTURE ONLY -- runs only in the build container, where
/root/reference is mounted.  It imports the reference (LinYark/Sdirt, pure
PyTorch) on CPU with stub modules, runs the PSF hot
path (deeplens/optics.py:916-996, deeplens/surfaces.py:391-830,
deeplens/monte_carlo.py:9-372, deeplens/render_psf.py:76-188) on small seeded
inputs and stores inputs, intermediate ray states and outputs as .npz data.
No reference source text is stored, only numbers.

Discipline: torch.set_num_threads(1); every case is generated
twice and asserted bit-equal; the non-deterministic paraxial pupil
(optics.py:1335-1376, lstsq on near-parallel lines: values drift run-to-run by
~1e-5 relative (entrance) / ~2e-4 (exit) even single-threaded) -- and hfov,
foclen and fnum, which the reference computes from fresh estimates -- are FROZEN
at the values oracle/frozen_lens_scalars.json records (one run of the reference
per lens, kept outside tests/golden/), so that a fresh run reproduces every
committed file byte for byte; the reference's fresh values are asserted to lie
within its own run-to-run spread of the frozen ones (oracle/ref_pupil_variation.py).
--refreeze draws new values from the reference (once per lens) instead -- every
fixture downstream then changes.

Usage:  python oracle/gen_golden.py [--out tests/golden] [--refreeze]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _refimport import import_reference  # noqa: E402

PSFNet, set_seed, deeplens = import_reference(num_threads=1)
from deeplens import surfaces as ref_surfaces  # noqa: E402
from deeplens import monte_carlo as ref_mc  # noqa: E402
from deeplens import optics as ref_optics  # noqa: E402
ref_render = sys.modules["deeplens.render_psf"]
from deeplens.basics import Ray, WAVE_RGB  # noqa: E402

DP_DEFAULT = [0.78, 1.44, 0.3, 0.5]  # h, f, w, r  (monte_carlo.py:157-164)


# --------------------------------------------------------------------------
# lens construction with frozen pupil
# --------------------------------------------------------------------------
COMMITTED = os.path.join(HERE, "..", "tests", "golden")
# One run of the reference per lens, kept OUTSIDE tests/golden/ (the fixtures are then reproduced FROM these numbers, not
# from themselves); provenance and the reference's own run-to-run spread are recorded in the file.
FROZEN_FILE = os.path.join(HERE, "frozen_lens_scalars.json")


def frozen_scalars(name):
    with open(FROZEN_FILE) as f:
        d = json.load(f)
    return d.get(name), d["_spread"]


def freeze(lens, name, refreeze=False):
    """Pin the run-to-run-unstable scalars of a reference lens (paraxial pupils, and hfov / foclen / fnum, which are
    computed from fresh pupil estimates: optics.py:1193-1196, 1203-1230) at the values of oracle/frozen_lens_scalars.json,
    after checking that this run's fresh values lie within the reference's own spread of them.  refreeze: record this
    run's values as the new frozen ones instead."""
    ent_z, ent_r = lens.calc_entrance_pupil_paraxial(entrance=True)
    ext_z, ext_r = lens.calc_entrance_pupil_paraxial(entrance=False)
    fresh = dict(pupil_z=float(ent_z), pupil_r=float(ent_r), exit_pupil_z=float(ext_z), exit_pupil_r=float(ext_r),
                 hfov=float(lens.hfov), foclen=float(lens.foclen), fnum=float(lens.fnum))
    st, spread = frozen_scalars(name)
    if refreeze or st is None:
        with open(FROZEN_FILE) as f:
            d = json.load(f)
        d[name] = st = fresh
        with open(FROZEN_FILE, "w") as f:
            json.dump(d, f, indent=1)
        print(f"{name}: froze this run's scalars {fresh}")
    tol = dict(pupil_z=spread["entrance"], pupil_r=spread["entrance"], exit_pupil_z=spread["exit"],
               exit_pupil_r=spread["exit"], hfov=spread["hfov"], foclen=spread["foclen"], fnum=spread["fnum"])
    for k, v in fresh.items():
        assert abs(v / st[k] - 1) <= tol[k], \
            f"{name}: the reference's fresh {k} = {v!r} is not within its run-to-run spread ({tol[k]:g}) of the frozen {st[k]!r}"
    drift = {k: f"{fresh[k] / st[k] - 1:+.1e}" for k in fresh if fresh[k] != st[k]}
    print(f"{name}: scalars frozen at oracle/frozen_lens_scalars.json (this run's relative drift: {drift or 'none'})")
    lens.hfov, lens.foclen, lens.fnum = st["hfov"], st["foclen"], st["fnum"]

    def frozen(M=32, entrance=True, shrink_pupil=False):
        z, r = (st["pupil_z"], st["pupil_r"]) if entrance else (st["exit_pupil_z"], st["exit_pupil_r"])
        if shrink_pupil:
            r = r * 0.25                         # optics.py:1394-1395
        return z, r
    lens.entrance_pupil = frozen
    return lens


def build_lens(name, refreeze=False):
    set_seed(0)
    lens = PSFNet(filename=f"/root/reference/lenses/{name}/lens_web.json",
                  sensor_res=(512, 768), kernel_size=21, device="cpu")
    lens.refocus(-1000 + lens.d_sensor)          # 1_fit_psfnet.py:23-25
    return freeze(lens, name, refreeze)


def lens_state(lens, wvlns):
    """Flat, JSON-able description of everything the hot path reads."""
    surfs = []
    for s in lens.surfaces:
        c = float(s.c.item())
        if c == 0.0:
            kind = "plane"
        elif s.ai is None and float(s.k.item()) == 0.0:
            kind = "sphere"
        else:
            kind = "asphere"
        ai = [float(getattr(s, f"ai{2 * i + 2}").item()) for i in range(s.ai_degree)]
        surfs.append(dict(
            kind=kind, r=float(s.r), d=float(s.d.item()), c=c, k=float(s.k.item()),
            ai=ai, mat1=s.mat1.name, mat2=s.mat2.name,
            n1={repr(w): float(s.mat1.ior(w)) for w in wvlns},
            n2={repr(w): float(s.mat2.ior(w)) for w in wvlns}))
    ez, er = lens.entrance_pupil()
    xz, xr = lens.entrance_pupil(entrance=False)
    return dict(
        lens_name=os.path.basename(os.path.dirname(lens.lens_name)),
        d_sensor=float(lens.d_sensor), hfov=float(lens.hfov),
        foclen=float(lens.foclen), fnum=float(lens.fnum),
        r_last=float(lens.r_last), sensor_size=[float(v) for v in lens.sensor_size],
        sensor_res=[int(v) for v in lens.sensor_res], pixel_size=float(lens.pixel_size),
        aper_idx=int(lens.aper_idx), pupil_z=float(ez), pupil_r=float(er),
        exit_pupil_z=float(xz), exit_pupil_r=float(xr), surfaces=surfs)


# --------------------------------------------------------------------------
# recording hooks
# --------------------------------------------------------------------------
class Recorder:
    """Monkeypatches the reference to capture RNG draws, per-surface ray
    states, Newton trip counts and raw L/R grids of one psf_diff call."""

    def __init__(self):
        self.rand = []          # list of np arrays, in draw order
        self.traces = []        # one dict per Lensgroup.trace call
        self.grids = []         # (l_grid, r_grid) per assign_points call
        self.sampled = []       # (o, d) after Ray.__init__ in sample_from_points
        self.centers = []
        self.pupil = []         # (x2, y2) as the reference computed them (optics.py:485-488)
        self._cur = None

    def __enter__(self):
        rec = self
        self._rand = torch.rand
        self._rr = ref_surfaces.Aspheric.ray_reaction
        self._vl = ref_surfaces.Aspheric._valid_loose
        self._tr = ref_optics.Lensgroup.trace
        self._sr = ref_mc.assign_points_to_pixels_small_r
        self._br = ref_mc.assign_points_to_pixels_big_r
        self._sfp = ref_optics.Lensgroup.sample_from_points
        self._pc = ref_optics.Lensgroup.psf_center
        self._stack = torch.stack

        def stack(tensors, *a, **k):
            # optics.py:488: o2 = torch.stack((x2, y2, z2), 1) -- the only 3-tuple of
            # 1-D tensors stacked along dim 1 on this path
            if (len(a) == 1 and a[0] == 1 and len(tensors) == 3
                    and all(t.dim() == 1 for t in tensors)):
                rec.pupil.append((tensors[0].numpy().copy(), tensors[1].numpy().copy()))
            return rec._stack(tensors, *a, **k)

        def rand(*a, **k):
            out = rec._rand(*a, **k)
            rec.rand.append(out.numpy().copy())
            return out

        def ray_reaction(self_, ray):
            rec._trips = 0
            out = rec._rr(self_, ray)
            if rec._cur is not None:
                rec._cur["o"].append(out.o.numpy().copy())
                rec._cur["d"].append(out.d.numpy().copy())
                rec._cur["ra"].append(out.ra.numpy().copy())
                rec._cur["obliq"].append(out.obliq.numpy().copy())
                rec._cur["trips"].append(rec._trips)
            return out

        def valid_loose(self_, x, y):
            rec._trips += 1
            return rec._vl(self_, x, y)

        def trace(self_, ray, lens_range=None, record=False):
            rec._cur = dict(o=[], d=[], ra=[], obliq=[], trips=[],
                            o_in=ray.o.numpy().copy(), d_in=ray.d.numpy().copy())
            out = rec._tr(self_, ray, lens_range=lens_range, record=record)
            rec.traces.append(rec._cur)
            rec._cur = None
            return out

        def small_r(*a, **k):
            l, r = rec._sr(*a, **k)
            rec.grids.append((l.numpy().copy(), r.numpy().copy()))
            return l, r

        def big_r(*a, **k):
            l, r = rec._br(*a, **k)
            rec.grids.append((l.numpy().copy(), r.numpy().copy()))
            return l, r

        def sfp(self_, *a, **k):
            ray = rec._sfp(self_, *a, **k)
            rec.sampled.append((ray.o.numpy().copy(), ray.d.numpy().copy()))
            return ray

        def pc(self_, *a, **k):
            c = rec._pc(self_, *a, **k)
            rec.centers.append(c.numpy().copy())
            return c

        torch.rand = rand
        torch.stack = stack
        ref_surfaces.Aspheric.ray_reaction = ray_reaction
        ref_surfaces.Aspheric._valid_loose = valid_loose
        ref_optics.Lensgroup.trace = trace
        ref_mc.assign_points_to_pixels_small_r = small_r
        ref_mc.assign_points_to_pixels_big_r = big_r
        ref_optics.assign_points_to_pixels_small_r = small_r
        ref_optics.assign_points_to_pixels_big_r = big_r
        ref_optics.Lensgroup.sample_from_points = sfp
        ref_optics.Lensgroup.psf_center = pc
        return self

    def __exit__(self, *exc):
        torch.rand = self._rand
        torch.stack = self._stack
        ref_surfaces.Aspheric.ray_reaction = self._rr
        ref_surfaces.Aspheric._valid_loose = self._vl
        ref_optics.Lensgroup.trace = self._tr
        ref_mc.assign_points_to_pixels_small_r = self._sr
        ref_mc.assign_points_to_pixels_big_r = self._br
        ref_optics.assign_points_to_pixels_small_r = self._sr
        ref_optics.assign_points_to_pixels_big_r = self._br
        ref_optics.Lensgroup.sample_from_points = self._sfp
        ref_optics.Lensgroup.psf_center = self._pc


# forward_integral looks the splat functions up in its own module globals
# (monte_carlo.py:60-62), so patching ref_mc.* is what takes effect.


def run_psf_case(lens, points, ks, spp, wvln, seed, param_list=None, full=True):
    """One psf_diff call under the recorder -> dict of arrays."""
    set_seed(seed)
    pts = torch.tensor(points, dtype=torch.float32)
    with Recorder() as rec:
        psf = lens.psf_diff(points=pts, wvln=wvln, ks=ks, spp=spp, center=True,
                            param_list=param_list)
    out = dict(points=pts.numpy(), ks=np.int32(ks), spp=np.int32(spp),
               wvln=np.float64(wvln), seed=np.int32(seed), psf=psf.numpy())
    assert len(rec.rand) == 4 and len(rec.traces) == 2
    out["u_theta"], out["u_r2"], out["uc_theta"], out["uc_r2"] = rec.rand
    out["center"] = rec.centers[0]
    assert len(rec.pupil) == 2
    (out["pupil_x2"], out["pupil_y2"]), (out["pupil_xc"], out["pupil_yc"]) = rec.pupil
    main, cen = rec.traces
    out["trips"] = np.asarray(main["trips"], np.int32)
    out["trips_center"] = np.asarray(cen["trips"], np.int32)
    out["grid_l"] = np.stack([g[0] for g in rec.grids])
    out["grid_r"] = np.stack([g[1] for g in rec.grids])
    if full:
        out["ray_o0"], out["ray_d0"] = rec.sampled[0]     # after normalise
        out["surf_o"] = np.stack(main["o"])               # [K,S,N,3]
        out["surf_d"] = np.stack(main["d"])
        out["surf_ra"] = np.stack(main["ra"])
        out["cen_o0"], out["cen_d0"] = rec.sampled[1]
        out["cen_o_last"] = cen["o"][-1]
        out["cen_d_last"] = cen["d"][-1]
        out["cen_ra_last"] = cen["ra"][-1]
    return out


def twice(fn):
    a = fn()
    b = fn()
    for k in a:
        va, vb = np.asarray(a[k]), np.asarray(b[k])
        assert va.shape == vb.shape and np.array_equal(va, vb, equal_nan=True), \
            f"fixture key {k} not reproducible"
    return a


def save(out_dir, name, d):
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **d)
    print(f"  wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


# --------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    ap.add_argument("--refreeze", action="store_true",
                    help="take new paraxial pupil values from the reference instead of the committed lens_state_*.json")
    args = ap.parse_args()
    out_dir = os.path.abspath(args.out)
    os.makedirs(out_dir, exist_ok=True)
    wv_all = [0.589] + list(WAVE_RGB)

    lenses = {}
    for name in ("rf50mm", "rf35mm"):
        lens = build_lens(name, args.refreeze)
        lenses[name] = lens
        st = lens_state(lens, wv_all)
        with open(os.path.join(out_dir, f"lens_state_{name}.json"), "w") as f:
            json.dump(st, f, indent=1)
        print(name, "d_sensor", st["d_sensor"], "hfov", st["hfov"], "pupil",
              st["pupil_z"], st["pupil_r"])

    rf50, rf35 = lenses["rf50mm"], lenses["rf35mm"]
    ds50, ds35 = rf50.d_sensor, rf35.d_sensor

    # F1 = BASELINE config 1 exactly (plumbing case)
    save(out_dir, "f1_rf50_c1", twice(lambda: run_psf_case(
        rf50, [[0.0, 0.0, -1500 + ds50]], ks=17, spp=256, wvln=0.589, seed=0)))

    # F2 = 4 points (axis near/far, two field corners), every checkpoint
    pts4 = [[0.0, 0.0, -300.0], [0.0, 0.0, -20000.0],
            [0.95, -0.9, -300.0], [-0.98, 0.98, -20000.0]]
    save(out_dir, "f2_rf50_pts4", twice(lambda: run_psf_case(
        rf50, pts4, ks=33, spp=64, wvln=0.589, seed=1)))
    # F2b = same rays, DP parameter list given -> non-zero R grid
    for direct in ("l", "r"):
        save(out_dir, f"f2_rf50_pts4_dp_{direct}", twice(lambda: run_psf_case(
            rf50, pts4, ks=33, spp=64, wvln=0.589, seed=1,
            param_list=DP_DEFAULT + [direct], full=False)))
    # big-r microlens branch (monte_carlo.py:242-372)
    save(out_dir, "f2_rf50_pts4_bigr", twice(lambda: run_psf_case(
        rf50, pts4, ks=33, spp=64, wvln=0.589, seed=1,
        param_list=[0.78, 1.44, 0.3, 0.6, "l"], full=False)))

    # F3 = second lens (21 surfaces)
    save(out_dir, "f3_rf35_pts4", twice(lambda: run_psf_case(
        rf35, pts4, ks=33, spp=64, wvln=0.589, seed=2)))

    # F4 = psf_rgb: three wavelengths, each a fresh psf_diff (optics.py:999-1015)
    def rgb():
        set_seed(3)
        pts = torch.tensor([[0.3, 0.2, -800.0], [-0.7, 0.6, -5000.0]])
        with Recorder() as rec:
            psf = rf50.psf_rgb(points=pts, ks=17, spp=64)
        d = dict(points=pts.numpy(), ks=np.int32(17), spp=np.int32(64),
                 seed=np.int32(3), psf=psf.numpy(),
                 wvlns=np.asarray(WAVE_RGB, np.float64))
        assert len(rec.rand) == 12
        d["rand"] = np.stack([r for r in rec.rand if r.shape[0] == 64])       # [6,64]
        d["rand_center"] = np.stack([r for r in rec.rand if r.shape[0] == 2048])
        d["centers"] = np.stack(rec.centers)
        d["pupil_x"] = np.stack([p[0] for p in rec.pupil if p[0].shape[0] == 64])     # [3,64]
        d["pupil_y"] = np.stack([p[1] for p in rec.pupil if p[0].shape[0] == 64])
        d["pupil_xc"] = np.stack([p[0] for p in rec.pupil if p[0].shape[0] == 2048])  # [3,2048]
        d["pupil_yc"] = np.stack([p[1] for p in rec.pupil if p[0].shape[0] == 2048])
        d["trips"] = np.stack([np.asarray(t["trips"], np.int32) for t in rec.traces])
        return d
    save(out_dir, "f4_rf50_rgb", twice(rgb))

    # F5 = splat only: synthetic (x, y, x_tan, ra) through both branches
    def splat():
        g = torch.Generator().manual_seed(5)
        S, ks, ps = 4096, 21, 0.046875
        rng = [(-ks / 2 + 0.5) * ps, (ks / 2 - 0.5) * ps]
        lim = rng[1] - 0.01 * ps
        pts = (torch.rand(S, 2, generator=g) * 2 - 1) * lim * 0.999
        x_tan = (torch.rand(S, generator=g) * 2 - 1) * 0.6
        ra = (torch.rand(S, generator=g) > 0.1).float()
        d = dict(points=pts.numpy(), x_tan=x_tan.numpy(), ra=ra.numpy(),
                 ks=np.int32(ks), ps=np.float64(ps))
        for tag, fn, pl in (("small", ref_mc.assign_points_to_pixels_small_r, DP_DEFAULT + ["l"]),
                            ("small_r04", ref_mc.assign_points_to_pixels_small_r, [0.7, 1.3, 0.25, 0.4, "l"]),
                            ("big", ref_mc.assign_points_to_pixels_big_r, [0.78, 1.44, 0.3, 0.6, "l"])):
            l, r = fn(points=pts.clone(), ks=ks, x_range=rng, y_range=rng, ra=ra.clone(),
                      x_tan=x_tan.clone(), param_list=pl)
            d[f"{tag}_l"], d[f"{tag}_r"] = l.numpy(), r.numpy()
            d[f"{tag}_param"] = np.asarray(pl[:4], np.float64)
        l, r = ref_mc.assign_points_to_pixels_small_r(
            points=pts.clone(), ks=ks, x_range=rng, y_range=rng, ra=ra.clone(),
            x_tan=x_tan.clone(), param_list=None)
        d["default_l"], d["default_r"] = l.numpy(), r.numpy()
        return d
    save(out_dir, "f5_splat", twice(splat))

    # F6 = forward_integral edge cases on a synthetic sensor-plane Ray:
    #      rays just inside/outside the window, dead rays, negative d_x.
    def edges():
        g = torch.Generator().manual_seed(6)
        S, N, ks, ps = 96, 3, 9, 0.046875
        half = (ks / 2 - 0.5 - 0.01) * ps
        o = torch.zeros(S, N, 3)
        o[..., :2] = (torch.rand(S, N, 2, generator=g) * 2 - 1) * half * 1.3
        # exact-boundary probes
        o[0, :, 0] = float(np.float32(half)); o[0, :, 1] = 0.0
        o[1, :, 0] = float(np.nextafter(np.float32(half), np.float32(0))); o[1, :, 1] = 0.0
        o[2, :, 1] = -float(np.float32(half)); o[2, :, 0] = 0.0
        o[..., 2] = 62.25
        dd = torch.randn(S, N, 3, generator=g) * 0.15
        dd[..., 2] = 1.0
        ra = (torch.rand(S, N, generator=g) > 0.2).float()
        cen = (torch.rand(N, 2, generator=g) - 0.5) * ps
        ray = Ray(o.clone(), dd.clone(), ra=ra.clone(), device="cpu")
        d = dict(o=o.numpy(), d=ray.d.numpy().copy(), ra=ra.numpy(), center=cen.numpy(),
                 ks=np.int32(ks), ps=np.float64(ps))
        with Recorder() as rec:
            psf = ref_mc.forward_integral(ray, ps=ps, ks=ks, pointc_ref=cen.clone(),
                                          param_list=DP_DEFAULT + ["l"])
        d["psf_l"] = psf.numpy()
        d["grid_l"] = np.stack([g_[0] for g_ in rec.grids])
        d["grid_r"] = np.stack([g_[1] for g_ in rec.grids])
        return d
    save(out_dir, "f6_window_edges", twice(edges))

    # F7 = image-space per-pixel PSF convolution (render_psf.py:76-188)
    def render():
        g = torch.Generator().manual_seed(7)
        B, C, H, W, ks = 2, 3, 8, 12, 5
        img = torch.rand(B, C, H, W, generator=g)
        psf = torch.rand(B, H, W, 2, ks, ks, generator=g)
        psf = psf / psf.sum((-1, -2), keepdim=True)
        d = dict(img=img.numpy(), psf=psf.numpy(), ks=np.int32(ks))
        rl, rr = ref_render.local_psf_render_fast(img.clone(), psf.clone(), kernel_size=ks)
        d["fast_l"], d["fast_r"] = rl.numpy(), rr.numpy()
        rl, rr = ref_render.local_psf_render(img.clone(), psf.clone(), kernel_size=ks)
        d["half_l"], d["half_r"] = rl.numpy(), rr.numpy()
        d["dp_fp32"] = ref_render.local_dp_psf_render(img.clone(), psf.clone(), kernel_size=ks).numpy()
        psf_g = torch.rand(C, ks, ks, generator=g)
        d["psf_global"] = psf_g.numpy()
        d["global"] = ref_render.render_psf(img.clone(), psf_g.clone()).numpy()
        return d
    save(out_dir, "f7_render", twice(render))

    # F8 = miniature BASELINE config 2 (3x3x3 volume, 4096 spp, ks 65): PSFs only
    def mini_c2():
        g = 3
        x, y = torch.meshgrid(torch.linspace(-1 + 1 / (2 * g), 1 - 1 / (2 * g), g),
                              torch.linspace(1 - 1 / (2 * g), -1 + 1 / (2 * g), g),
                              indexing="xy")
        z = rf50.z2depth(torch.linspace(0, 1, g))
        pts = torch.stack([x.reshape(-1, 1).expand(-1, g).reshape(-1),
                           y.reshape(-1, 1).expand(-1, g).reshape(-1),
                           z.repeat(g * g)], -1)
        d = run_psf_case(rf50, pts.tolist(), ks=65, spp=4096, wvln=0.589, seed=8,
                         param_list=DP_DEFAULT + ["l"], full=False)
        d.pop("grid_l")          # keep raw R (needed), drop raw L (= psf * max)
        return d
    save(out_dir, "f8_rf50_mini_c2", twice(mini_c2))
    d = run_psf_case(rf50, twice(mini_c2)["points"].tolist(), ks=65, spp=4096, wvln=0.589,
                     seed=8, param_list=DP_DEFAULT + ["r"], full=False)
    save(out_dir, "f8_rf50_mini_c2_r", dict(psf=d["psf"]))


if __name__ == "__main__":
    main()
