#!/usr/bin/env python3
"""How far is the reference from ITSELF?  Same seed, same inputs, 1 CPU thread vs 8 (the mini
config-2 volume of fixture f8: 27 points, 4096 spp, ks 65).  Build container only (imports the
reference).  Prints max / median |dPSF| relative to the PSF peak (= 1 after max-normalisation)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _refimport import import_reference  # noqa: E402

PSFNet, set_seed, _ = import_reference(num_threads=1)


def run(threads):
    torch.set_num_threads(threads)
    set_seed(0)
    lens = PSFNet(filename="/root/reference/lenses/rf50mm/lens_web.json", sensor_res=(512, 768),
                  kernel_size=21, device="cpu")
    lens.refocus(-1000 + lens.d_sensor)
    g = 3
    x, y = torch.meshgrid(torch.linspace(-1 + 1 / (2 * g), 1 - 1 / (2 * g), g),
                          torch.linspace(1 - 1 / (2 * g), -1 + 1 / (2 * g), g), indexing="xy")
    z = lens.z2depth(torch.linspace(0, 1, g))
    pts = torch.stack([x.reshape(-1, 1).expand(-1, g).reshape(-1), y.reshape(-1, 1).expand(-1, g).reshape(-1),
                       z.repeat(g * g)], -1)
    set_seed(8)
    return lens.psf(points=pts, ks=65, spp=4096).numpy(), lens.entrance_pupil()


a, pa = run(1)
b, pb = run(8)
c, pc = run(1)
for name, (u, v) in {"run 1 (1 thread) vs run 2 (8 threads)": (a, b), "run 1 (1 thread) vs run 3 (1 thread)": (a, c),
                     "run 2 (8 threads) vs run 3 (1 thread)": (b, c)}.items():
    d = np.abs(u - v).reshape(len(u), -1)
    print(f"{name}: max {d.max():.2e}  median of per-PSF max {np.median(d.max(1)):.2e}  "
          f"fraction of PSFs changed {np.mean(d.max(1) > 0):.2f}")
print("entrance pupil (z, r):", pa, pb, pc)
