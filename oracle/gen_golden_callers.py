#!/usr/bin/env python3
"""Fixtures from the reference's plotting callers of the PSF path (SURVEY.md §8b "Callers").

F24 = Lensgroup.draw_mtf (deeplens/optics.py:2041-2067) run as it stands on rf50mm: three fields
      (0, 0.7, 1.0 of the diagonal) at DEPTH, each `psf_diff(point, wvln, ks=256)` -- the one caller
      that asks for grids larger than a workgroup's LDS -- followed by psf2mtf.  Recorded: the three
      256x256 PSFs, their pupil sample sets (hand-off), centres, trip tables and the three
      (freq, tangential, sagittal) curves the plot draws.
F25 = Lensgroup.draw_psf_radial (optics.py:1934-1956): M = 3 fields along the 45-degree diagonal,
      `psf_rgb(point, ks=51, center=True, spp=4096)` divided by its maximum (and its log-scaled
      form); the list handed to make_grid is recorded (make_grid / save_image themselves are
      torchvision, absent here: the recorder returns a placeholder image).

TEST INFRASTRUCTURE ONLY -- build container only (imports /root/reference).
"""
import argparse
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402

import matplotlib  # noqa: E402
matplotlib.use("Agg")


def mtf_plot(rf50):
    calls = []
    orig = gg.ref_optics.Lensgroup.psf2mtf

    def psf2mtf(self_, psf, diag=False):
        out = orig(self_, psf, diag)
        calls.append((psf.numpy().copy(), [np.asarray(v) for v in out]))
        return out

    gg.set_seed(24)
    gg.ref_optics.Lensgroup.psf2mtf = psf2mtf
    try:
        with gg.Recorder() as rec, tempfile.TemporaryDirectory() as tmp:
            rf50.draw_mtf(save_name=os.path.join(tmp, "mtf.png"))
    finally:
        gg.ref_optics.Lensgroup.psf2mtf = orig
    assert len(calls) == 3 and len(rec.pupil) == 6 and len(rec.traces) == 6 and len(rec.centers) == 3
    assert calls[0][0].shape == (256, 256)
    return dict(relative_fov=np.asarray([0.0, 0.7, 1.0], np.float32), depth=np.float32(-20000.0),
                ks=np.int32(256), spp=np.int32(2048), seed=np.int32(24), wvln=np.float64(0.589),
                psf=np.stack([c[0] for c in calls]),
                freq=np.stack([c[1][0] for c in calls]), tangential=np.stack([c[1][1] for c in calls]),
                sagittal=np.stack([c[1][2] for c in calls]),
                pupil_x=np.stack([rec.pupil[2 * i][0] for i in range(3)]),
                pupil_y=np.stack([rec.pupil[2 * i][1] for i in range(3)]),
                pupil_xc=np.stack([rec.pupil[2 * i + 1][0] for i in range(3)]),
                pupil_yc=np.stack([rec.pupil[2 * i + 1][1] for i in range(3)]),
                center=np.stack([c.reshape(2) for c in rec.centers]),
                trips=np.stack([np.asarray(rec.traces[2 * i]["trips"], np.int32) for i in range(3)]),
                trips_center=np.stack([np.asarray(rec.traces[2 * i + 1]["trips"], np.int32) for i in range(3)]),
                pixel_size=np.float64(rf50.pixel_size), hfov=np.float64(rf50.hfov))


def radial_plot(rf50, log_scale):
    handed = []

    def make_grid(psfs, **kw):
        handed.append(([p.numpy().copy() for p in psfs], dict(kw)))
        return torch.zeros(3, 4, 4)

    def save_image(*a, **k):
        return None

    gg.set_seed(25)
    old = gg.ref_optics.make_grid, gg.ref_optics.save_image
    gg.ref_optics.make_grid, gg.ref_optics.save_image = make_grid, save_image
    try:
        with gg.Recorder() as rec:
            rf50.draw_psf_radial(M=3, ks=51, log_scale=log_scale, save_name="unused.png")
    finally:
        gg.ref_optics.make_grid, gg.ref_optics.save_image = old
    assert len(handed) == 1 and len(handed[0][0]) == 3 and len(rec.pupil) == 18
    assert handed[0][1] == dict(nrow=3, padding=1, pad_value=0.0)
    # per point: three wavelengths, each (primary set, chief-ray set)
    # -> [point, wavelength, (x, y), sample]
    pup = lambda j: np.stack([np.stack([np.stack(rec.pupil[6 * i + 2 * w + j]) for w in range(3)]) for i in range(3)])
    return dict(psfs=np.stack(handed[0][0]), pupil=pup(0), pupil_c=pup(1),
                M=np.int32(3), ks=np.int32(51), spp=np.int32(4096), seed=np.int32(25),
                depth=np.float32(-20000.0), log_scale=np.bool_(log_scale))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    out = os.path.abspath(ap.parse_args().out)
    rf50 = gg.build_lens("rf50mm")
    gg.save(out, "f24_rf50_draw_mtf", gg.twice(lambda: mtf_plot(rf50)))
    lin, log = gg.twice(lambda: radial_plot(rf50, False)), gg.twice(lambda: radial_plot(rf50, True))
    assert np.array_equal(lin["pupil"], log["pupil"])
    lin["psfs_log"] = log["psfs"]
    gg.save(out, "f25_rf50_draw_psf_radial", lin)
