#!/usr/bin/env python3
"""One training iteration of 2_dfdp_net.py at its own settings (configs/dfdp_by_sdirt_rf50mm.yml:
bs 4, 512x768, ks 21, n_stack 1) on synthetic RGB-D: simulate the DP pairs (PSFNet.render with the
training noise, one image at a time as the reference does, 2_dfdp_net.py:166-171), then the depth
network forward + backward + AdamW under fp16 autocast with loss scaling and gradient clipping."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdirt_amd.dfdp import Basenet
from sdirt_amd.psfnet import PSFNet

dev = "cuda:0"
torch.backends.cudnn.benchmark = os.environ.get("SDIRT_MIOPEN_FIND", "0") == "1"
bs, H, W, ks = 4, 512, 768, 21
torch.manual_seed(0); np.random.seed(0)
lens = PSFNet(os.path.join(os.path.dirname(__file__), "..", "sdirt_amd", "data", "rf50mm.json"),
              sensor_res=(H, W), kernel_size=ks, device=dev)
lens.refocus(-1000 + lens.d_sensor)
with torch.no_grad():
    lens.psfnet.net[-2].bias.add_(0.02)
net = Basenet("dfdp").to(dev).train()
optim = torch.optim.AdamW(net.parameters(), 1e-4)
scaler = torch.amp.GradScaler("cuda")
g = torch.Generator(device=dev).manual_seed(0)
aif = torch.rand(bs, 3, H, W, device=dev, generator=g)
gt_depth = 0.5 + 4.5 * torch.rand(bs, 1, H, W, device=dev, generator=g)           # metres


def iteration():
    with torch.no_grad():
        foc = 0.5 + 4.5 * torch.rand(bs, device=dev)
        stack = torch.cat([lens.render(aif[i:i + 1], depth=-gt_depth[i:i + 1] * 1e3, foc_dist=-foc[i:i + 1] * 1e3,
                                       train=True) for i in range(bs)], dim=0)
    t_render = time.perf_counter()
    optim.zero_grad()
    losses, _ = net({"gt_depth": gt_depth.clone(), "AiF_img": aif, "stack_rgb_img": stack})
    loss = losses["total"].mean()
    scaler.scale(loss).backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=1.0)
    scaler.step(optim)
    scaler.update()
    return loss, t_render


for _ in range(3):
    iteration()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
tr = tn = 0.0
for _ in range(n):
    e0.record()
    with torch.no_grad():
        foc = 0.5 + 4.5 * torch.rand(bs, device=dev)
        stack = torch.cat([lens.render(aif[i:i + 1], depth=-gt_depth[i:i + 1] * 1e3, foc_dist=-foc[i:i + 1] * 1e3,
                                       train=True) for i in range(bs)], dim=0)
    e1.record()
    optim.zero_grad()
    losses, _ = net({"gt_depth": gt_depth.clone(), "AiF_img": aif, "stack_rgb_img": stack})
    scaler.scale(losses["total"].mean()).backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), max_norm=1.0)
    scaler.step(optim)
    scaler.update()
    e2.record()
    torch.cuda.synchronize()
    tr += e0.elapsed_time(e1); tn += e1.elapsed_time(e2)
dt = (time.perf_counter() - t0) / n * 1e3
print(f"2_dfdp_net.py iteration (bs {bs}, {H}x{W}): {dt:.1f} ms = DP-pair simulation {tr / n:.1f} ms "
      f"({tr / n / bs:.2f} ms per image) + depth-network step {tn / n:.1f} ms; loss {float(losses['total']):.4f}")
