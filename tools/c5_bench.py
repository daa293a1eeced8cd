#!/usr/bin/env python3
"""BASELINE config 5 (image simulation): PSFNet.render on a synthetic RGB-D frame, 512x768,
ks 21, full-size MLP (3 -> 128 -> 512 x9 -> 441) with seeded random weights (the reference's
checkpoints are not in its repository).  Times the stages with HIP events."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdirt_amd.psfnet import PSFNet
from sdirt_amd.render_psf import local_psf_render_fast

dev = "cuda:0"
torch.backends.cudnn.benchmark = os.environ.get("SDIRT_MIOPEN_FIND", "0") == "1"
H, W, ks = 512, 768, 21
torch.manual_seed(0)
m = PSFNet(os.path.join(os.path.dirname(__file__), "..", "sdirt_amd", "data", "rf50mm.json"),
           sensor_res=(H, W), kernel_size=ks, device=dev)
m.refocus(-1000 + m.d_sensor)
with torch.no_grad():
    m.psfnet.net[-2].bias.add_(0.02)
g = torch.Generator(device=dev).manual_seed(0)
img = torch.rand(1, 3, H, W, device=dev, generator=g)
depth = -(500 + 4500 * torch.rand(1, 1, H, W, device=dev, generator=g))
foc = torch.tensor([-1000.0], device=dev)


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


x, y = torch.meshgrid(torch.linspace(-1, 1, W), torch.linspace(1, -1, H), indexing="xy")
o = torch.stack((x.to(dev)[None], y.to(dev)[None], m.depth2z(depth + m.d_sensor).squeeze(1)), -1).float()
with torch.no_grad():
    t_net = timed(lambda: m.psfnet(o))
    t_pred = timed(lambda: m.pred(o.clone()))
    psf = m.pred(o.clone())
    t_conv = timed(lambda: local_psf_render_fast(img, psf, ks))
    t_all = timed(lambda: m.render(img, depth, foc))
    t_fmlp = timed(lambda: m.psfnet.forward_fused(o, mirror=True))
    m.fused_mlp = False
    t_gemm = timed(lambda: m.render(img, depth, foc))
    m.fused_render = False
    t_chain = timed(lambda: m.render(img, depth, foc))
macs = 3 * 128 + 128 * 512 + 8 * 512 * 512 + 512 * ks * ks
fl = 2 * macs * H * W
print(f"psf dtype {psf.dtype}; one network pass {t_net:.2f} ms ({fl / t_net / 1e9:.0f} TFLOP/s); "
      f"pred (L+R, normalise) {t_pred:.2f} ms; convolution {t_conv:.2f} ms; fused MLP both passes {t_fmlp:.2f} ms ({2 * fl / t_fmlp / 1e9:.0f} TFLOP/s); "
      f"render total {t_all:.2f} ms (torch.nn layers + fused conv {t_gemm:.2f} ms, op-by-op chain {t_chain:.2f} ms) "
      f"-> {1e3 / t_all:.1f} frames/s")

# ---- second half of config 5: depth-from-DP network forward on the simulated pair ----------
from sdirt_amd.dfdp import DfDPNet, dp_cost_volume
torch.manual_seed(1)
net = DfDPNet().to(dev).eval()
with torch.no_grad():
    pair = m.render(img, depth, foc)
    left, right = pair[:, :3].contiguous(), pair[:, 3:].contiguous()
    t32 = timed(lambda: net(left, right), n=5, warm=2)
    with torch.autocast("cuda", dtype=torch.float16):
        t16 = timed(lambda: net(left, right), n=5, warm=2)
        f = net.feature(left)
    t_cv = timed(lambda: dp_cost_volume(f, f, 20))

    def ref_cv(x, y, d_max=20):
        B, C, H_, W_ = x.shape
        cost = torch.zeros(B, C * 2, d_max, H_, W_).type_as(x)
        for i in range(d_max):
            gap = i - d_max // 2
            keep = slice(None, gap) if gap < 0 else slice(gap, None)
            cost[:, :C, i, :, keep] = x[:, :, :, keep]
            cost[:, C:, i, :, keep] = y[:, :, :, -gap:] if gap < 0 else (y[:, :, :, :-gap] if gap > 0 else y)
        return cost
    t_cv_ref = timed(lambda: ref_cv(f, f))
print(f"DfDP net forward 512x768: fp32 {t32:.1f} ms, fp16 autocast {t16:.1f} ms; cost volume "
      f"[1,64,20,128,192] {f.dtype}: HIP kernel {t_cv * 1e3:.0f} us vs zero-fill + 40 slice copies {t_cv_ref * 1e3:.0f} us; "
      f"config 5 end to end (render + depth net, fp16) {t_all + t16:.1f} ms")
