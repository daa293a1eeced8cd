#!/usr/bin/env python3
"""Golden fixtures for the PSF-network consumers of the PSF path:
PSFNet.init_net / pred / pred_coc / render / degamma / gamma / train_psfnet
(deeplens/psfnet.py:63-168, 317-379, 589-714) and the MLP it wraps
(deeplens/psfnet_arch.py:26-50, 291-303).

TEST INFRASTRUCTURE ONLY -- runs only in the build container (imports the
reference from /root/reference on CPU, one thread).  Stores numbers only:
inputs, seeded weights of a SMALL network of the reference's architecture class
and the reference's outputs.  Every case is generated twice and asserted equal.

Usage:  python oracle/gen_golden_psfnet.py [--out tests/golden]
"""
import argparse
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _refimport import import_reference  # noqa: E402
from gen_golden import twice, save  # noqa: E402  (also imports the reference)

PSFNet, set_seed, deeplens = import_reference(num_threads=1)
ref_arch = sys.modules["deeplens.psfnet_arch"]
ref_psfnet = sys.modules["deeplens.psfnet"]

KS = 7
HIDDEN, LAYERS = 32, 2


def small_net(seed):
    torch.manual_seed(seed)
    net = ref_arch.MLP(in_features=3, out_features=KS ** 2, hidden_features=HIDDEN,
                       hidden_layers=LAYERS)
    net.apply(ref_arch.initialize_weights)           # psfnet.py:89
    # kaiming + zero bias + final ReLU leaves ~half of the outputs dead; shift the last bias so
    # that every tap of the kernel is exercised
    with torch.no_grad():
        net.net[-2].bias.add_(0.05)
    return net


def make_psfnet():
    set_seed(0)
    m = PSFNet(filename="/root/reference/lenses/rf50mm/lens_web.json", sensor_res=(512, 768),
               kernel_size=KS, device="cpu")
    m.refocus(-1000 + m.d_sensor)
    return m


def weights(net, prefix="w/"):
    return {prefix + k: v.detach().numpy().copy() for k, v in net.state_dict().items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    out_dir = os.path.abspath(ap.parse_args().out)
    model = make_psfnet()

    # F9a: init_net reproducibility -- checksums of the full-size network (psfnet.py:78) built
    # right after manual_seed(0), as init_net does (constructor init + its own apply + apply)
    def init_sums():
        torch.manual_seed(0)
        model.kernel_size, model.model_name = 21, "mlp"
        model.init_net()
        d = {}
        for k, v in model.psfnet.state_dict().items():
            d["sum/" + k] = np.float64(v.double().sum().item())
            d["abs/" + k] = np.float64(v.double().abs().sum().item())
            d["head/" + k] = v.reshape(-1)[:4].numpy().copy()
            d["shape/" + k] = np.asarray(v.shape, np.int64)
        model.kernel_size = KS
        return d
    save(out_dir, "f9_psfnet_init", twice(init_sums))

    # F9b: pred / pred_coc / render / tone curves with a small seeded MLP
    def forward_cases():
        model.kernel_size = KS
        model.psfnet = small_net(9)
        g = torch.Generator().manual_seed(10)
        d = dict(ks=np.int32(KS), hidden=np.int32(HIDDEN), layers=np.int32(LAYERS),
                 d_sensor=np.float64(model.d_sensor), foclen=np.float64(model.foclen),
                 fnum=np.float64(model.fnum))
        d.update(weights(model.psfnet))
        inp = torch.rand(1, 4, 6, 3, generator=g)
        inp[..., :2] = inp[..., :2] * 2 - 1
        d["pred_inp"] = inp.numpy().copy()
        with torch.no_grad():
            d["pred"] = model.pred(inp.clone()).numpy()
            d["pred_coc"] = model.pred_coc(inp.clone()).numpy()
        B, C, H, W = 2, 3, 8, 12
        img = torch.rand(B, C, H, W, generator=g)
        depth = -(200 + torch.rand(B, 1, H, W, generator=g) * 4000)
        foc = torch.tensor([-1000.0, -700.0])
        d["img"], d["depth"], d["foc_dist"] = img.numpy(), depth.numpy(), foc.numpy()
        d["render"] = model.render(img.clone(), depth.clone(), foc.clone()).numpy()
        x = torch.rand(64, generator=g)
        d["tone_in"] = x.numpy()
        d["degamma"] = model.degamma(x.clone()).numpy()
        d["gamma"] = model.gamma(model.degamma(x.clone())).numpy()
        return d
    save(out_dir, "f9_psfnet_forward", twice(forward_cases))

    # F9c: four optimiser steps of train_psfnet (psfnet.py:101-168) on fixed batches
    def train_case():
        model.kernel_size = KS
        model.psfnet = small_net(11)
        g = torch.Generator().manual_seed(12)
        bs, iters = 8, 3
        batches = []
        for _ in range(iters + 1):
            inp = torch.rand(bs, 3, generator=g)
            psf = torch.rand(bs, KS, KS, generator=g)
            psf = psf / psf.amax((-1, -2), keepdim=True)
            batches.append((inp, psf))
        d = dict(ks=np.int32(KS), hidden=np.int32(HIDDEN), layers=np.int32(LAYERS),
                 lr=np.float64(1e-3), iters=np.int32(iters),
                 inp=np.stack([b[0].numpy() for b in batches]),
                 psf=np.stack([b[1].numpy() for b in batches]))
        d.update(weights(model.psfnet, "w0/"))
        feed = iter(batches)
        orig = model.get_training_data
        model.get_training_data = lambda bs, spp: tuple(t.clone() for t in next(feed))
        try:
            with tempfile.TemporaryDirectory() as tmp:
                model.train_psfnet(iters=iters, bs=bs, lr=1e-3, spp=16, evaluate_every=10 ** 6,
                                   result_dir=tmp)
        finally:
            model.get_training_data = orig
        d.update(weights(model.psfnet, "w1/"))
        return d
    save(out_dir, "f9_psfnet_train", twice(train_case))


if __name__ == "__main__":
    main()
