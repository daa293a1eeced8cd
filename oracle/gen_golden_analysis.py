#!/usr/bin/env python3
"""Fixture F26: the lens report of the reference's fitting script (1_fit_psfnet.py:29-32 ->
Lensgroup.analysis, deeplens/optics.py:1663-1684) on rf50mm at the script's near depth (-500 mm + d_sensor).

  * the three ray fans of the layout figure (plot_setup2D_with_trace, optics.py:1722-1738: views 0, 0.707 and
    0.99 of the half field in blue / green / red light, 9 rays each from sample_point_source_2D through the
    entrance pupil): rays as sampled, recorded paths `oss` of trace2sensor(record=True), final weights;
  * calc_magnification3 / calc_scale_ray (optics.py:1237-1321) and analysis_rms (optics.py:2103-2140) under a
    seed (the reference draws the per-source pupil samples with torch.rand on its device: the CPU generator
    here), with the first sample plane of the magnification bundle for a check of the sampler itself;
  * the lens title string and the 35-mm-equivalent focal length.

TEST INFRASTRUCTURE ONLY -- build container only (imports /root/reference).
"""
import argparse
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
from gen_golden_boundary import pad  # noqa: E402

import matplotlib  # noqa: E402
matplotlib.use("Agg")


def layout(rf50, depth):
    fans = []
    orig = gg.ref_optics.Lensgroup.trace2sensor

    def trace2sensor(self_, ray, record=False, ignore_invalid=False):
        o, d = ray.o.numpy().copy(), ray.d.numpy().copy()
        out = orig(self_, ray, record=record, ignore_invalid=ignore_invalid)
        if record:
            fans.append((o, d, float(ray.wvln), out[1], ray.ra.numpy().copy()))
        return out

    titles = []
    orig_title = matplotlib.axes.Axes.set_title

    def set_title(self_, label, *a, **k):
        titles.append(label)
        return orig_title(self_, label, *a, **k)

    gg.ref_optics.Lensgroup.trace2sensor = trace2sensor
    matplotlib.axes.Axes.set_title = set_title
    try:
        with tempfile.TemporaryDirectory() as tmp:
            rf50.plot_setup2D_with_trace(filename=os.path.join(tmp, "layout"), entrance_pupil=True, depth=depth)
            assert os.path.getsize(os.path.join(tmp, "layout.png")) > 10000
    finally:
        gg.ref_optics.Lensgroup.trace2sensor = orig
        matplotlib.axes.Axes.set_title = orig_title
    assert len(fans) == 3 and len(titles) == 1
    layout.title = titles[0]
    out = dict(eqfl=np.float64(rf50.calc_eqfl()), fnum=np.float64(rf50.fnum),
               foclen=np.float64(rf50.foclen), aper_idx=np.int32(rf50.aper_idx))
    for i, (o, d, w, oss, ra) in enumerate(fans):
        out[f"fan{i}_o"], out[f"fan{i}_d"], out[f"fan{i}_wvln"], out[f"fan{i}_ra"] = o, d, np.float64(w), ra
        out[f"fan{i}_len"], out[f"fan{i}_pts"] = pad(oss)
    return out


def measures(rf50, depth):
    planes = []
    orig = gg.ref_optics.Lensgroup.sample_point_source

    def sample_point_source(self_, *a, **k):
        ray = orig(self_, *a, **k)
        planes.append((ray.o[0].numpy().copy(), ray.d[0].numpy().copy(), tuple(ray.o.shape)))
        return ray

    gg.ref_optics.Lensgroup.sample_point_source = sample_point_source
    try:
        gg.set_seed(26)
        mag = rf50.calc_magnification3(depth)
        gg.set_seed(26)
        scale = rf50.calc_scale_ray(depth)
        gg.set_seed(27)
        rms = [float(v) for v in rf50.analysis_rms(depth=depth)]
        gg.set_seed(27)
        rms_own = [float(v) for v in rf50.analysis_rms(depth=depth, ref=False)]
    finally:
        gg.ref_optics.Lensgroup.sample_point_source = orig
    assert planes[0][2] == (512, 21, 21, 3) and planes[2][2] == (512, 21, 21, 3) and planes[3][2] == (2048, 31, 31, 3)
    return dict(mag=np.float64(mag), scale=np.float64(scale), rms=np.asarray(rms), rms_own=np.asarray(rms_own),
                mag_o0=planes[0][0], mag_d0=planes[0][1], rms_o0=planes[3][0], rms_d0=planes[3][1],
                seed_mag=np.int32(26), seed_rms=np.int32(27))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    out = os.path.abspath(ap.parse_args().out)
    rf50 = gg.build_lens("rf50mm")
    depth = -500 + rf50.d_sensor                                   # 1_fit_psfnet.py:28
    d = gg.twice(lambda: layout(rf50, depth))
    d.update(gg.twice(lambda: measures(rf50, depth)))
    d["depth"] = np.float64(depth)
    d["title"] = np.asarray(layout.title)
    gg.save(out, "f26_rf50_analysis", d)
