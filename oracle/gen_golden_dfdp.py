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
    # Basenet.dfdp (dfdp/basenet.py:23-49) in depth-estimation mode, same seed: its first member is
    # the same YRStereonet_3D, so the weights above are drawn again
    bspec = importlib.util.spec_from_file_location("dfdp_ref.basenet", "/root/reference/dfdp/basenet.py",
                                                   submodule_search_locations=None)
    pkg = types.ModuleType("dfdp_ref"); pkg.__path__ = ["/root/reference/dfdp"]; sys.modules["dfdp_ref"] = pkg
    dd = types.ModuleType("dfdp_ref.dddnet"); dd.__path__ = []; sys.modules["dfdp_ref.dddnet"] = dd
    sys.modules["dfdp_ref.dddnet.dddnet"] = ref
    bn = importlib.util.module_from_spec(bspec); bn.__package__ = "dfdp_ref"
    bspec.loader.exec_module(bn)
    torch.manual_seed(21)
    base = bn.Basenet(train_mode="dfdp").eval()
    gt = 0.5 + 4.5 * torch.rand(1, 1, 128, 128, generator=g)
    gt[0, 0, :4, :4] = 0.0                                  # invalid ground truth -> masked out
    d["gt_depth_sum"] = np.float64(gt.double().sum().item())
    with torch.no_grad():
        losses, outputs = base.dfdp({"stack_rgb_img": torch.cat((xl, xr), 1), "AiF_img": xl,
                                     "gt_depth": gt.clone()}, train=True)
    d["loss_total"] = np.float64(losses["total"].item())
    d["pred_depth_est"] = outputs["pred_depth_est"].numpy()
    d["gt_depth_roundtrip_err"] = np.float64((outputs["gt_depth"] - gt).abs().max().item())
    # train_mode='deblur' (basenet.py:29-31, 65-69): the Mydeblur branch on top
    torch.manual_seed(21)
    base = bn.Basenet(train_mode="deblur").eval()
    for k, v in base.deblur_net.state_dict().items():
        d["dsum/" + k] = np.float64(v.double().sum().item())
        d["dabs/" + k] = np.float64(v.double().abs().sum().item())
    with torch.no_grad():
        base.deblur_net.cam_attention.gamma.fill_(0.3)        # exercise the attention branch
        losses, outputs = base.dfdp({"stack_rgb_img": torch.cat((xl, xr), 1), "AiF_img": xl,
                                     "gt_depth": gt.clone()}, train=True)
    d["deblur_loss_total"] = np.float64(losses["total"].item())
    d["deblur_losses"] = np.asarray([losses["depth_est"].item(), losses["depth_fix"].item(),
                                     losses["aif"].item()], np.float64)
    d["pred_aif_head"] = outputs["pred_aif"][0, :, ::16, ::16].numpy()
    d["pred_depth_fix_head"] = outputs["pred_depth_fix"][0, :, ::16, ::16].numpy()
    return d


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    save(os.path.abspath(ap.parse_args().out), "f10_dfdp_net", twice(case))
