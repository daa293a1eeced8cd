#!/usr/bin/env python3
"""Fixture F14: the reference's own RAYS of the miniature config-2 volume (tests/golden/f8: 3x3x3
points, 4096 spp, ks 65, seed 8), for the ray-hand-off parity test.

TEST INFRASTRUCTURE ONLY -- build container only (imports /root/reference).  Stores numbers, no
reference source: the post-normalise directions of the primary and the chief-ray pass (`ray.d`
as Ray.__init__ leaves it, deeplens/basics.py:245), the object-space points, the Newton trip
counts of both passes, the chief-ray centres, the max-normalised L PSF and the raw R grid.  With
the rays handed over, everything upstream of the trace (pupil estimate, disc mapping, the
cancelling subtraction d = o2 - o) is out of the comparison: what remains is trace -> propagate ->
splat -> normalise, the part SURVEY.md §7 sets the 1e-5 bar for.

A second run of the same call (`*_cr` arrays) executes the reference with its elementary functions
made CORRECTLY ROUNDED inside trace and splat: torch.sqrt / acos / sin / cos evaluated in float64
and rounded once (torch's fp32 CPU kernels for them are MKL VML: < 1 ulp, not correctly rounded),
`r2 ** n` for n >= 4 as the float64 product chain rounded once.  Nothing else changes -- same
reference code, same rays, same operation order.  This isolates how much of the distance between the
reference and an IEEE-exact evaluation of its own operation sequence is the math library's last bit.

Usage:  python oracle/gen_golden_handoff.py [--out tests/golden]
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (imports the reference, one thread)


class CorrectlyRoundedMath:
    """While active, torch.sqrt / acos / asin / sin / cos of fp32 tensors and tensor ** n (n >= 4)
    are evaluated through float64 and rounded once."""
    NAMES = ("sqrt", "acos", "asin", "sin", "cos")

    def __enter__(self):
        self.saved = {n: getattr(torch, n) for n in self.NAMES}
        self.pow = torch.Tensor.__pow__

        def via64(fn):
            def f(x, *a, **k):
                if torch.is_tensor(x) and x.dtype == torch.float32:
                    return fn(x.double(), *a, **k).float()
                return fn(x, *a, **k)
            return f
        for n, fn in self.saved.items():
            setattr(torch, n, via64(fn))
        pow0 = self.pow

        def tpow(x, n):
            if torch.is_tensor(x) and x.dtype == torch.float32 and isinstance(n, int) and n >= 4:
                acc = x.double()
                p = acc
                for _ in range(n - 1):
                    acc = acc * p
                return acc.float()
            return pow0(x, n)
        torch.Tensor.__pow__ = tpow
        return self

    def __exit__(self, *exc):
        for n, fn in self.saved.items():
            setattr(torch, n, fn)
        torch.Tensor.__pow__ = self.pow


class CorrectlyRoundedInsideTraceAndSplat:
    """Activates CorrectlyRoundedMath inside Lensgroup.trace and forward_integral only, so that the
    sampled rays stay the ones of the plain run."""

    def __enter__(self):
        self.tr = gg.ref_optics.Lensgroup.trace
        self.fi = gg.ref_optics.forward_integral
        tr, fi = self.tr, self.fi

        def trace(self_, *a, **k):
            with CorrectlyRoundedMath():
                return tr(self_, *a, **k)

        def fwd(*a, **k):
            with CorrectlyRoundedMath():
                return fi(*a, **k)
        gg.ref_optics.Lensgroup.trace = trace
        gg.ref_optics.forward_integral = fwd
        return self

    def __exit__(self, *exc):
        gg.ref_optics.Lensgroup.trace = self.tr
        gg.ref_optics.forward_integral = self.fi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    args = ap.parse_args()
    rf50 = gg.build_lens("rf50mm")
    # (gen_golden.build_lens has frozen the run-to-run-unstable pupils at oracle/frozen_lens_scalars.json.)
    # Prerequisites -- the lens state and fixture f8, both written by gen_golden.py -- from --out when this run's
    # gen_golden.py wrote them there (check_regenerable.py: a clean directory, generators in order), else the committed ones
    import json

    def prerequisite(name):
        p = os.path.join(args.out, name)
        return p if os.path.exists(p) else os.path.join(gg.COMMITTED, name)
    with open(prerequisite("lens_state_rf50mm.json")) as f:
        st = json.load(f)
    assert st["d_sensor"] == float(rf50.d_sensor) and st["hfov"] == float(rf50.hfov)
    assert (st["pupil_z"], st["pupil_r"]) == tuple(rf50.entrance_pupil())
    f8 = np.load(prerequisite("f8_rf50_mini_c2.npz"))

    def case():
        d = gg.run_psf_case(rf50, f8["points"].tolist(), ks=65, spp=4096, wvln=0.589, seed=8,
                            param_list=gg.DP_DEFAULT + ["l"], full=True)
        return dict(points=d["points"], point_obj=d["ray_o0"][0], ray_d0=d["ray_d0"], cen_d0=d["cen_d0"],
                    trips=d["trips"], trips_center=d["trips_center"], center=d["center"], psf=d["psf"],
                    grid_r=d["grid_r"], ks=d["ks"], spp=d["spp"], seed=d["seed"])
    d = gg.twice(case)
    # the same call as fixture f8: the PSFs must be the ones already committed
    assert np.array_equal(d["psf"], f8["psf"]) and np.array_equal(d["grid_r"], f8["grid_r"])

    def case_cr():
        with CorrectlyRoundedInsideTraceAndSplat():
            return case()
    c = gg.twice(case_cr)
    assert np.array_equal(c["ray_d0"], d["ray_d0"]) and np.array_equal(c["cen_d0"], d["cen_d0"])
    for k in ("trips", "trips_center", "center", "psf", "grid_r"):
        d[k + "_cr"] = c[k]
    print("plain vs correctly-rounded-math reference: trips equal", np.array_equal(c["trips"], d["trips"]),
          " max |dPSF_L| %.2e" % np.abs(c["psf"] - d["psf"]).max(),
          " max |dcentre| %.2e" % np.abs(c["center"] - d["center"]).max())
    gg.save(os.path.abspath(args.out), "f14_rf50_mini_c2_rays", d)

    # F20 = the same hand-off on rf35mm (21 surfaces, stop at index 7, an even asphere): 12 of the
    # 27 points, 4096 spp, ks 65, with the correctly-rounded-math re-run
    rf35 = gg.build_lens("rf35mm")
    with open(prerequisite("lens_state_rf35mm.json")) as f:
        st35 = json.load(f)
    assert st35["d_sensor"] == float(rf35.d_sensor) and st35["hfov"] == float(rf35.hfov)
    assert (st35["pupil_z"], st35["pupil_r"]) == tuple(rf35.entrance_pupil())
    pts35 = f8["points"][::2][:12].tolist()

    def case35():
        d_ = gg.run_psf_case(rf35, pts35, ks=65, spp=4096, wvln=0.589, seed=20,
                             param_list=gg.DP_DEFAULT + ["l"], full=True)
        return dict(points=d_["points"], point_obj=d_["ray_o0"][0], ray_d0=d_["ray_d0"], cen_d0=d_["cen_d0"],
                    trips=d_["trips"], trips_center=d_["trips_center"], center=d_["center"], psf=d_["psf"],
                    grid_r=d_["grid_r"], ks=d_["ks"], spp=d_["spp"], seed=d_["seed"])
    d35 = gg.twice(case35)

    def case35_cr():
        with CorrectlyRoundedInsideTraceAndSplat():
            return case35()
    c35 = gg.twice(case35_cr)
    assert np.array_equal(c35["ray_d0"], d35["ray_d0"])
    for k in ("trips", "trips_center", "center", "psf", "grid_r"):
        d35[k + "_cr"] = c35[k]
    print("rf35mm, plain vs correctly-rounded-math reference: trips equal", np.array_equal(c35["trips"], d35["trips"]),
          " max |dPSF_L| %.2e" % np.abs(c35["psf"] - d35["psf"]).max())
    gg.save(os.path.abspath(args.out), "f20_rf35_handoff_rays", d35)

    # F15 = forward_integral WITHOUT a reference centre (pointc_ref=None -> the RMS centre of the
    # rays themselves, monte_carlo.py:27-31) on synthetic sensor-plane rays, both DP outputs
    def rms_centre():
        g = torch.Generator().manual_seed(15)
        S, N, ks, ps = 512, 5, 17, 0.046875
        o = torch.zeros(S, N, 3)
        spread = torch.tensor([0.02, 0.05, 0.1, 0.2, 0.3]).reshape(1, N, 1)
        o[..., :2] = torch.randn(S, N, 2, generator=g) * spread + (torch.rand(1, N, 2, generator=g) - 0.5)
        o[..., 2] = 62.25
        dd = torch.randn(S, N, 3, generator=g) * 0.15
        dd[..., 2] = 1.0
        ra = (torch.rand(S, N, generator=g) > 0.15).float()
        out = dict(o=o.numpy(), ra=ra.numpy(), ks=np.int32(ks), ps=np.float64(ps))
        for direct in ("l", "r"):
            ray = gg.Ray(o.clone(), dd.clone(), ra=ra.clone(), device="cpu")
            out["d"] = ray.d.numpy().copy()
            with gg.Recorder() as rec:
                psf = gg.ref_mc.forward_integral(ray, ps=ps, ks=ks, pointc_ref=None,
                                                 param_list=gg.DP_DEFAULT + [direct])
            out[f"psf_{direct}"] = psf.numpy()
            out[f"grid_l_{direct}"] = np.stack([g_[0] for g_ in rec.grids])
            out[f"grid_r_{direct}"] = np.stack([g_[1] for g_ in rec.grids])
        ray = gg.Ray(o.clone(), dd.clone(), ra=ra.clone(), device="cpu")
        out["psf_default"] = gg.ref_mc.forward_integral(ray, ps=ps, ks=ks).numpy()
        pts = -o[..., :2]
        out["rms_center"] = ((pts * ra.unsqueeze(-1)).sum(0) / ra.unsqueeze(-1).sum(0).add(1e-9)).numpy()
        return out
    gg.save(os.path.abspath(args.out), "f15_rms_center", gg.twice(rms_centre))

    # F16 = psf_diff(center=False): PSFs centred on the ideal (pinhole) image point instead of the
    # chief ray (optics.py:972-976); only TWO random vectors are drawn
    def uncentred():
        pts = [[0.0, 0.0, -300.0], [0.3, -0.2, -1000.0], [0.95, -0.9, -300.0], [-0.98, 0.98, -20000.0]]
        gg.set_seed(16)
        with gg.Recorder() as rec:
            psf = rf50.psf_diff(points=torch.tensor(pts), wvln=0.589, ks=33, spp=1024, center=False,
                                param_list=gg.DP_DEFAULT + ["l"])
        assert len(rec.rand) == 2 and len(rec.traces) == 1 and len(rec.pupil) == 1
        return dict(points=np.asarray(pts, np.float32), ks=np.int32(33), spp=np.int32(1024), seed=np.int32(16),
                    psf=psf.numpy(), u_theta=rec.rand[0], u_r2=rec.rand[1], pupil_x2=rec.pupil[0][0],
                    pupil_y2=rec.pupil[0][1], trips=np.asarray(rec.traces[0]["trips"], np.int32),
                    grid_l=np.stack([g_[0] for g_ in rec.grids]), grid_r=np.stack([g_[1] for g_ in rec.grids]))
    gg.save(os.path.abspath(args.out), "f16_rf50_uncentred", gg.twice(uncentred))

    # F17 = psf_rgb at the sampling density of config 2 (4096 spp) on the 3 x 3 field psf_map uses
    # (point_source_grid, optics.py:816-861), with every pupil sample set for ray-level hand-off.
    # psf_map itself (optics.py:1018-1041) is this tensor tiled by torchvision's make_grid, which the
    # container lacks; the fixture holds the tiles.
    def rgb_field():
        field = rf50.point_source_grid(depth=-1500.0, grid=3)
        gg.set_seed(17)
        with gg.Recorder() as rec:
            psf = rf50.psf_rgb(points=field.reshape(-1, 3), ks=33, spp=4096)
        assert len(rec.rand) == 12 and len(rec.pupil) == 6 and len(rec.traces) == 6
        prim = [p for p in rec.pupil if p[0].shape[0] == 4096]
        cent = [p for p in rec.pupil if p[0].shape[0] == 2048]
        return dict(field=field.numpy(), depth=np.float64(-1500.0), grid=np.int32(3), ks=np.int32(33),
                    spp=np.int32(4096), seed=np.int32(17), psf=psf.numpy(),
                    pupil_x=np.stack([p[0] for p in prim]), pupil_y=np.stack([p[1] for p in prim]),
                    pupil_xc=np.stack([p[0] for p in cent]), pupil_yc=np.stack([p[1] for p in cent]),
                    trips=np.stack([np.asarray(t["trips"], np.int32) for t in rec.traces[0::2]]),
                    trips_center=np.stack([np.asarray(t["trips"], np.int32) for t in rec.traces[1::2]]),
                    centers=np.stack(rec.centers), wvlns=np.asarray(gg.WAVE_RGB, np.float64))
    gg.save(os.path.abspath(args.out), "f17_rf50_rgb_field", gg.twice(rgb_field))

    # F18 = local_psf_render_high_res (render_psf.py:191-208): the image cut into patches, each
    # rendered on its own by local_psf_render (so every patch is replicate-padded at ITS border).
    # The reference's function itself raises (it assigns local_psf_render's (left, right) tuple into
    # one tensor); the fixture is what its loop computes when both halves are kept: the reference's
    # own local_psf_render applied patch by patch.
    def high_res():
        g = torch.Generator().manual_seed(18)
        B, C, H, W, ks, patch = 2, 3, 10, 14, 5, (4, 6)
        img = torch.rand(B, C, H, W, generator=g)
        psf = torch.rand(B, H, W, 2, ks, ks, generator=g)
        psf = psf / psf.sum((-1, -2), keepdim=True)
        rl, rr = torch.zeros_like(img), torch.zeros_like(img)
        for i0 in range(0, H, patch[0]):
            for j0 in range(0, W, patch[1]):
                i1, j1 = min(i0 + patch[0], H), min(j0 + patch[1], W)
                a, b = gg.ref_render.local_psf_render(img[:, :, i0:i1, j0:j1].clone(),
                                                      psf[:, i0:i1, j0:j1].clone(), kernel_size=ks)
                rl[:, :, i0:i1, j0:j1], rr[:, :, i0:i1, j0:j1] = a, b
        return dict(img=img.numpy(), psf=psf.numpy(), ks=np.int32(ks), patch=np.asarray(patch, np.int32),
                    left=rl.numpy(), right=rr.numpy())
    gg.save(os.path.abspath(args.out), "f18_render_high_res", gg.twice(high_res))

    # F19 = point_source_grid (optics.py:816-861) for every option combination the shim mirrors
    def grids():
        out = {}
        for grid in (1, 2, 5, 8):
            for center in (False, True):
                for quater in (False, True):
                    for normalized in (True, False):
                        if grid == 1 and quater:
                            continue
                        p = rf50.point_source_grid(depth=-1234.5, grid=grid, normalized=normalized,
                                                   quater=quater, center=center)
                        out[f"g{grid}_c{int(center)}_q{int(quater)}_n{int(normalized)}"] = p.numpy()
        return out
    gg.save(os.path.abspath(args.out), "f19_point_source_grid", gg.twice(grids))


if __name__ == "__main__":
    main()
