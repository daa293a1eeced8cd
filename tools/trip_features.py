#!/usr/bin/env python3
"""Is the verified Newton trip table of a PSFNet training batch predictable from cheap batch
features (extreme field radius / depth)?  Prints, per distinct table, the feature ranges."""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdirt_amd.psfnet import PSFNet

torch.manual_seed(0); np.random.seed(0)
m = PSFNet(os.path.join(os.path.dirname(__file__), "..", "sdirt_amd", "data", "rf50mm.json"),
           sensor_res=(512, 768), kernel_size=21, device="cuda:0")
m.refocus(-1000 + m.d_sensor)
rows = []
orig_psf = m.psf


def psf(points, **kw):
    out = orig_psf(points=points, **kw)
    p = points.numpy()
    r = np.hypot(p[:, 0], p[:, 1])
    rows.append((tuple(int(x) for x in m.trips.cache[("center", "lean")]),
                 tuple(int(x) for x in m.trips.cache[("psf", 0.589, "lean")]),
                 r.max(), np.abs(p[:, 0]).max(), np.abs(p[:, 1]).max(), p[:, 2].max(), p[:, 2].min(),
                 (r * (1 / -p[:, 2])).max(), (r ** 2 / -p[:, 2]).max()))
    return out


m.psf = psf
for _ in range(400):
    m.get_training_data(bs=64, spp=20000)
names = ["rmax", "|x|max", "|y|max", "zmax(nearest)", "zmin(farthest)", "max r/|z|", "max r^2/|z|"]
for which, col in (("center", 0), ("psf", 1)):
    groups = collections.defaultdict(list)
    for row in rows:
        groups[row[col]].append(row[2:])
    print(which)
    for tab, vals in sorted(groups.items(), key=lambda kv: -len(kv[1])):
        v = np.asarray(vals)
        print("  ", "".join(format(x, "x") for x in tab), len(v),
              " ".join(f"{n}=[{v[:, i].min():.3g},{v[:, i].max():.3g}]" for i, n in enumerate(names)))
