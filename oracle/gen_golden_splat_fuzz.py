#!/usr/bin/env python3
"""Fixture F13: the reference's two splat branches (monte_carlo.py:135-372) on random dual-pixel
geometries (h, f, w, r) -- six parameter sets on both sides of r = 0.5, 1024 rays each.
TEST INFRASTRUCTURE ONLY -- build container only."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402


def case():
    rng = np.random.default_rng(13)
    g = torch.Generator().manual_seed(13)
    S, ks, ps = 1024, 21, 0.046875
    xr = [(-ks / 2 + 0.5) * ps, (ks / 2 - 0.5) * ps]
    lim = xr[1] - 0.01 * ps
    d = dict(ks=np.int32(ks), ps=np.float64(ps))
    params = []
    for i in range(6):
        h = float(rng.uniform(0.4, 1.1)); f = h + float(rng.uniform(0.3, 1.2))
        w = float(rng.uniform(0.1, 0.45)); r = float(rng.uniform(0.15, 0.48) if i % 2 == 0 else rng.uniform(0.52, 0.95))
        params.append([h, f, w, r])
        pts = (torch.rand(S, 2, generator=g) * 2 - 1) * lim * 0.999
        x_tan = (torch.rand(S, generator=g) * 2 - 1) * 0.7
        ra = (torch.rand(S, generator=g) > 0.1).float()
        fn = gg.ref_mc.assign_points_to_pixels_small_r if r <= 0.5 else gg.ref_mc.assign_points_to_pixels_big_r
        l, rr = fn(points=pts.clone(), ks=ks, x_range=xr, y_range=xr, ra=ra.clone(), x_tan=x_tan.clone(),
                   param_list=[h, f, w, r, "l"])
        d[f"points{i}"], d[f"x_tan{i}"], d[f"ra{i}"] = pts.numpy(), x_tan.numpy(), ra.numpy()
        d[f"l{i}"], d[f"r{i}"] = l.numpy(), rr.numpy()
    d["params"] = np.asarray(params, np.float64)
    return d


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    gg.save(os.path.abspath(ap.parse_args().out), "f13_splat_fuzz", gg.twice(case))
