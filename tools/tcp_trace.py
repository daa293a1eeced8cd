#!/usr/bin/env python3
"""Twelve back-to-back `to_host(psf(host points))` spans of the reference's timing harness, for a rocprofv3 timeline
(rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --output-format csv -d <dir> -- python3 tools/tcp_trace.py):
prints each span's wall time and monotonic start/end stamps."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sdirt_amd.psfnet import PSFNet  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = PSFNet(os.path.join(ROOT, "sdirt_amd", "data", "rf50mm.json"), sensor_res=(512, 768), kernel_size=21, device=dev)
m.refocus(-1000 + m.d_sensor)
inp = torch.rand(24576, 3)
inp[:, 2] = m.z2depth(inp[:, 2])
for _ in range(4):
    m.to_host(m.psf(points=inp, ks=21, spp=4096))
for i in range(12):
    torch.cuda.synchronize()
    t0 = time.monotonic_ns()
    m.to_host(m.psf(points=inp, ks=21, spp=4096))
    t1 = time.monotonic_ns()
    print(f"span {i}: {(t1 - t0) / 1e6:8.3f} ms  [{t0} .. {t1}]", flush=True)
