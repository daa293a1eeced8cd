#!/usr/bin/env python3
"""A/B of the depth network's forward pass (config 5's second half, dddnet.py:122-152) at 512 x 768, fp16 autocast, after
MIOpen's find pass -- interleaved rounds, HIP-event medians:

  two_calls   feature(left), feature(right) one after the other, planar (NCHW / NCDHW) tensors: the reference's order of
              calls (DfDPNet.inference_layout = False)
  product     DfDPNet in eval mode as shipped: ONE feature pass over both views, channels_last feature network, the cost
              volume written pixel-major (sdirt_dp_cost_volume_nhwc), channels_last_3d hourglass

The variants this came from (one call over both views with planar tensors; channels_last for the 2-D network only; a
re-laying copy in front of a channels_last_3d hourglass) are in profiles/r06/dfdp_ab_variants.txt.

  python tools/dfdp_ab.py [rounds]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdirt_amd.dfdp import DfDPNet  # noqa: E402

dev = torch.device("cuda", 0)
torch.backends.cudnn.benchmark = True
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = torch.Generator(device=dev).manual_seed(0)
left = torch.rand(1, 3, 512, 768, device=dev, generator=g)
right = torch.rand(1, 3, 512, 768, device=dev, generator=g)
nets = {}
for name, flag in (("two_calls", False), ("product", True)):
    torch.manual_seed(1)
    nets[name] = DfDPNet().to(dev).eval()
    nets[name].inference_layout = flag


def timed(fn, n=20):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    outs = {}
    for name, net in nets.items():
        for _ in range(3):
            outs[name] = net(left, right).float()
        torch.cuda.synchronize()
    print(f"max |disp(product) - disp(two_calls)| = {float((outs['product'] - outs['two_calls']).abs().max()):.3e} "
          f"(range of disp {float(outs['two_calls'].min()):.3f} .. {float(outs['two_calls'].max()):.3f})", flush=True)
    for r in range(rounds):
        print(f"round {r}: " + "  ".join(f"{name} {timed(lambda: net(left, right)):.3f} ms" for name, net in nets.items()), flush=True)
