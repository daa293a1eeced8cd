#!/usr/bin/env python3
"""Golden fixture for the depth-from-DP network: YRStereonet_3D
(dfdp/dddnet/dddnet.py:103-152) run on the CPU with seeded weights.

TEST INFRASTRUCTURE ONLY -- build container only.  The reference hard-codes
torch.cuda.current_device() in DisparityRegression (dddnet.py:564); the two torch.cuda hooks it
touches are replaced by CPU stand-ins for the duration of this script.  Stored: the inputs, the
seed, checksums of every parameter (the build's module must draw the same weights from the same
seed), the cost volume of a small feature pair, and the network's outputs.
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _refimport import import_reference  # noqa: E402
from gen_golden import twice, save  # noqa: E402

import_reference(num_threads=1)
sys.modules["skimage"].io = types.ModuleType("skimage.io")
sys.modules["skimage.io"] = sys.modules["skimage"].io
spec = importlib.util.spec_from_file_location("ref_dddnet", "/root/reference/dfdp/dddnet/dddnet.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


class _NoDevice:
    def __init__(self, *_):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


torch.cuda.current_device = lambda: "cpu"
torch.cuda.device_of = _NoDevice


def case():
    torch.manual_seed(21)
    net = ref.YRStereonet_3D().eval()
    g = torch.Generator().manual_seed(22)
    xl = torch.rand(1, 3, 128, 128, generator=g)
    xr = torch.roll(xl, 3, dims=-1) * 0.9 + 0.05 * torch.rand(1, 3, 128, 128, generator=g)
    # the inputs are re-drawn from these seeds by the test (torch's CPU generator is portable)
    d = dict(seed=np.int32(21), input_seed=np.int32(22), xl_sum=np.float64(xl.double().sum().item()),
             xr_sum=np.float64(xr.double().sum().item()))
    for k, v in net.state_dict().items():
        if v.dtype.is_floating_point:
            d["sum/" + k] = np.float64(v.double().sum().item())
            d["abs/" + k] = np.float64(v.double().abs().sum().item())
    with torch.no_grad():
        fl, fr = net.feature(xl), net.feature(xr)
        d["feature_l_head"] = fl[0, :, ::8, ::8].numpy()
        d["feature_l_abs"] = np.float64(fl.double().abs().sum().item())
        fx, fy = torch.rand(2, 3, 4, 24, generator=g), torch.rand(2, 3, 4, 24, generator=g)
        d["cv_x"], d["cv_y"] = fx.numpy(), fy.numpy()
        d["cv"] = ref.YRStereonet_3D.get_dp_cost_volume(fx, fy, 20).numpy()
        d["disp"] = net(xl, xr).numpy()
    return d


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    save(os.path.abspath(ap.parse_args().out), "f10_dfdp_net", twice(case))
