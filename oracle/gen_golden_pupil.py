#!/usr/bin/env python3
"""Fixture F12: the two 16-ray traces behind the reference's paraxial pupils
(deeplens/optics.py:1335-1361) -- BACKWARD from the stop through the front group (entrance pupil)
and forward from the stop through the rear group (exit pupil) -- with every per-surface state.
The only backward tracing on the path; pins `Lensgroup.trace(lens_range=...)` in both directions.

TEST INFRASTRUCTURE ONLY -- build container only.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


def case(name):
    gg.set_seed(0)
    lens = gg.PSFNet(filename=f"/root/reference/lenses/{name}/lens_web.json", sensor_res=(512, 768),
                     kernel_size=21, device="cpu")
    d = dict(aper_idx=np.int32(lens.aper_idx))
    for tag, entrance in (("ent", True), ("ext", False)):
        with gg.Recorder() as rec:
            lens.calc_entrance_pupil_paraxial(entrance=entrance)
        assert len(rec.traces) == 1
        tr = rec.traces[0]
        d[tag + "_o_in"], d[tag + "_d_in"] = tr["o_in"], tr["d_in"]
        d[tag + "_o"], d[tag + "_d"] = np.stack(tr["o"]), np.stack(tr["d"])      # traversal order
        d[tag + "_ra"] = np.stack(tr["ra"])
        d[tag + "_trips"] = np.asarray(tr["trips"], np.int32)
    return d


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    out = os.path.abspath(ap.parse_args().out)
    for name in ("rf50mm", "rf35mm"):
        gg.save(out, f"f12_pupil_traces_{name}", gg.twice(lambda: case(name)))
