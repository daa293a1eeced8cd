#!/usr/bin/env python3
"""What the chip does under k_psfnet_mlp: the kernel in a loop for ~6 s while `rocm-smi` is sampled from a child process
(power, shader clock) -- is the 1.88 GHz the kernel runs at the power cap?   python tools/mlp_power.py"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdirt_amd.psfnet import PSFNet  # noqa: E402

dev = "cuda:0"
H, W, ks = 512, 768, 21
torch.manual_seed(0)
m = PSFNet(os.path.join(os.path.dirname(__file__), "..", "sdirt_amd", "data", "rf50mm.json"), sensor_res=(H, W), kernel_size=ks, device=dev)
o = torch.rand(1, H, W, 3, device=dev) * 2 - 1
samples = []


def sample():
    for _ in range(5):
        time.sleep(1.0)
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=20).stdout
            keep = [ln.strip() for ln in out.splitlines() if any(k in ln for k in ("Power", "sclk", "Max Graphics"))]
            samples.append(" | ".join(keep))
        except Exception as e:      # noqa: BLE001
            samples.append(f"rocm-smi: {e}")


with torch.no_grad():
    m.psfnet.forward_fused(o, mirror=True)
    torch.cuda.synchronize()
    th = threading.Thread(target=sample)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < 6.5:
        for _ in range(20):
            m.psfnet.forward_fused(o, mirror=True)
        torch.cuda.synchronize()
        n += 20
    dt = time.time() - t0
    th.join()
print(f"k_psfnet_mlp x {n}: {dt / n * 1e3:.3f} ms per call over {dt:.1f} s")
for s in samples:
    print(s)
