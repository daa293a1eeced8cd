#!/usr/bin/env python3
"""How often does the speculated Newton trip table of one PSF call hold for the next one?
Prints the distinct verified tables over random PSFNet training batches."""
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
seen = collections.Counter()
for i in range(300):
    m.get_training_data(bs=64, spp=20000)
    seen[tuple((k if isinstance(k, str) else str(k), tuple(int(x) for x in v)) for k, v in sorted(m.trips.cache.items(), key=lambda kv: str(kv[0])))] += 1
for tabs, n in seen.most_common(12):
    print(n, *[f"{k}:{''.join(format(x, 'x') for x in v)}" for k, v in tabs])
print("rounds checked", m.trips.launches, "| corrected and re-rendered on the device", m.trips.device_relaunches,
      "| re-launched by the host", m.trips.relaunches)
