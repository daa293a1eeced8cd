#!/usr/bin/env python3
"""Fixtures for the two boundary leftovers of round 2.

F21 = Lensgroup.trace(ray, record=True) / trace2sensor(ray, record=True) -> `oss`, the per-ray
      list of intersection points the reference's plotting code walks (deeplens/optics.py:601-689):
      a fan of 11 meridional rays through rf50mm, forward, three of them vignetted on the way,
      and a backward fan from the sensor.  Stored padded: oss_len[i] points of ray i in oss_pts[i].
F22 = psf_rgb(points, center=False) (optics.py:999-1015 with :972-976): three wavelengths, PSFs
      centred on the pinhole image point, only TWO random vectors per wavelength; pupil sample
      sets recorded for the hand-off.

TEST INFRASTRUCTURE ONLY -- build container only (imports /root/reference).
"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


def pad(oss):
    n = max(len(o) for o in oss)
    pts = np.full((len(oss), n, 3), np.nan, np.float32)
    for i, o in enumerate(oss):
        pts[i, :len(o)] = np.stack([np.asarray(p, np.float32) for p in o])
    return np.asarray([len(o) for o in oss], np.int32), pts


def recorded(rf50):
    out = {}
    # forward fan: object point 800 mm in front, aimed across (and beyond) the first aperture
    n = 11
    o = torch.tensor([[0.0, 6.0, -800.0]]).repeat(n, 1)
    aim = torch.zeros(n, 3)
    aim[:, 1] = torch.linspace(-1.25, 1.25, n) * float(rf50.surfaces[0].r)
    aim[:, 2] = float(rf50.surfaces[0].d)
    out["fwd_o"], out["fwd_aim"] = o.numpy(), aim.numpy()
    ray = gg.Ray(o.clone(), aim - o, device="cpu")
    out["fwd_d"] = ray.d.numpy().copy()
    ray, valid, oss = rf50.trace(ray, record=True)
    out["fwd_valid"] = valid.numpy()
    out["fwd_len"], out["fwd_pts"] = pad(oss)
    ray = gg.Ray(o.clone(), aim - o, device="cpu")
    p, oss = rf50.trace2sensor(ray, record=True)
    out["sensor_p"] = p.numpy()
    out["sensor_len"], out["sensor_pts"] = pad(oss)
    # backward fan: from a sensor point towards the rear aperture
    o = torch.tensor([[0.0, -4.0, float(rf50.d_sensor)]]).repeat(n, 1)
    aim = torch.zeros(n, 3)
    aim[:, 1] = torch.linspace(-1.1, 1.1, n) * float(rf50.surfaces[-1].r)
    aim[:, 2] = float(rf50.surfaces[-1].d)
    out["bwd_o"], out["bwd_aim"] = o.numpy(), aim.numpy()
    ray = gg.Ray(o.clone(), aim - o, device="cpu")
    ray, valid, oss = rf50.trace(ray, record=True)
    out["bwd_valid"] = valid.numpy()
    out["bwd_len"], out["bwd_pts"] = pad(oss)
    out["d_sensor"] = np.float64(rf50.d_sensor)
    return out


def rgb_uncentred(rf50):
    pts = [[0.0, 0.0, -500.0], [0.4, -0.3, -1500.0], [-0.9, 0.85, -6000.0]]
    gg.set_seed(22)
    with gg.Recorder() as rec:
        psf = rf50.psf_rgb(points=torch.tensor(pts), ks=33, spp=2048, center=False,
                           param_list=gg.DP_DEFAULT + ["l"])
    assert len(rec.rand) == 6 and len(rec.pupil) == 3 and len(rec.traces) == 3
    gg.set_seed(22)
    psf_r = rf50.psf_rgb(points=torch.tensor(pts), ks=33, spp=2048, center=False,
                         param_list=gg.DP_DEFAULT + ["r"])
    return dict(points=np.asarray(pts, np.float32), ks=np.int32(33), spp=np.int32(2048), seed=np.int32(22),
                psf=psf.numpy(), psf_r=psf_r.numpy(),
                pupil_x=np.stack([p[0] for p in rec.pupil]), pupil_y=np.stack([p[1] for p in rec.pupil]),
                trips=np.stack([np.asarray(t["trips"], np.int32) for t in rec.traces]),
                wvlns=np.asarray(gg.WAVE_RGB, np.float64))


def mtf_case(rf50):
    """F23 = Lensgroup.psf2mtf (optics.py:1043-1080), the FFT consumer of a PSF behind draw_mtf: a seeded
    31x31 kernel -> (freq, tangential, sagittal)."""
    g = torch.Generator().manual_seed(23)
    yy, xx = torch.meshgrid(torch.arange(31.0), torch.arange(31.0), indexing="ij")
    psf = torch.exp(-((xx - 15.3) ** 2 / 18 + (yy - 14.6) ** 2 / 7)) + 0.02 * torch.rand(31, 31, generator=g)
    psf = psf / psf.max()
    freq, tan, sag = rf50.psf2mtf(psf)
    return dict(psf=psf.numpy(), freq=np.asarray(freq), tangential=np.asarray(tan), sagittal=np.asarray(sag),
                pixel_size=np.float64(rf50.pixel_size))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    out = os.path.abspath(ap.parse_args().out)
    rf50 = gg.build_lens("rf50mm")
    gg.save(out, "f21_rf50_recorded_paths", gg.twice(lambda: recorded(rf50)))
    gg.save(out, "f22_rf50_rgb_uncentred", gg.twice(lambda: rgb_uncentred(rf50)))
    gg.save(out, "f23_psf2mtf", gg.twice(lambda: mtf_case(rf50)))
