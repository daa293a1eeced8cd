"""Network architectures that represent the PSF field (consumers of the ray-traced PSFs).

Parameter names and shapes follow deeplens/psfnet_arch.py so that the reference's
checkpoints (`F4_PSFNet_mlp.pkl`: keys `net.<2i>.weight/bias`) load unchanged:
  MLP      psfnet_arch.py:26-50   in -> h/4 -> h -> (h -> h) x L -> out, ReLU after every layer
  MLPConv  psfnet_arch.py:76-136  MLP encoder to a (ks//4)^2 seed, transposed-conv decoder
The layers are stock torch.nn (hipBLASLt / MIOpen GEMMs on ROCm); the fp16 autocast the
reference wraps MLP.forward in (psfnet_arch.py:46) is applied on CUDA devices only.
"""
import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as nnF

from . import _lib
from .basics import dptr, stream_ptr


@torch.no_grad()
def initialize_weights(m):
    """psfnet_arch.py:291-303.  In-place initialisers on the parameters themselves (not on
    `.data`): the writes bump the tensors' version counters, which the packed-weight cache of the
    fused MLP kernel keys on."""
    if isinstance(m, nn.Conv2d):
        nn.init.kaiming_uniform_(m.weight, nonlinearity="relu")
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.BatchNorm2d):
        nn.init.constant_(m.weight, 1)
        nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.Linear):
        nn.init.kaiming_uniform_(m.weight)
        nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.ConvTranspose2d):
        nn.init.xavier_uniform_(m.weight)
    if hasattr(m, "invalidate_packed"):
        m.invalidate_packed()


def _autocast_for(t):
    return torch.autocast("cuda", dtype=torch.float16, enabled=t.is_cuda)


class MLP(nn.Module):
    def __init__(self, in_features, out_features, hidden_features=64, hidden_layers=3):
        super().__init__()
        self.ks = int(out_features ** (1 / 2))
        widths = [in_features, hidden_features // 4] + [hidden_features] * (hidden_layers + 1)
        layers = []
        for a, b in zip(widths[:-1], widths[1:]):
            layers += [nn.Linear(a, b, bias=True), nn.ReLU(inplace=True)]
        layers += [nn.Linear(hidden_features, out_features, bias=True), nn.ReLU()]
        self.net = nn.Sequential(*layers)
        self.net.apply(initialize_weights)

    def forward(self, inp):
        with _autocast_for(inp):
            x = self.net(inp)
        return x.reshape(*x.shape[:-1], self.ks, self.ks)

    # ---- inference in one HIP kernel (sdirt_psfnet_mlp) --------------------------------------
    def _linears(self):
        return [m for m in self.net if isinstance(m, nn.Linear)]

    def fused_supported(self):
        """Shapes sdirt_psfnet_mlp is built for: 3 -> h4 -> 512 -> ... -> 512 -> out."""
        lin = self._linears()
        widths = [lin[0].in_features] + [m.out_features for m in lin]
        return (len(lin) >= 3 and widths[0] == 3 and widths[1] in (32, 64, 96, 128)
                and all(w == 512 for w in widths[2:-1]) and 1 <= widths[-1] <= 512
                and all(m.bias is not None for m in lin))

    def invalidate_packed(self):
        """Forget the packed fp16 weight fragments of the fused kernel.  Needed only after writing
        parameters THROUGH `.data` (which does not bump their version counters): optimiser steps,
        load_state_dict, in-place initialisers and .to() are noticed by themselves."""
        self._pack_cache = None

    def _packed(self):
        """Weights as fp16 MFMA fragments + fp32 biases in one device buffer, rebuilt whenever a
        parameter was written (optimiser step, load_state_dict, in-place init) or moved; see
        invalidate_packed() for writes through `.data`."""
        lin = self._linears()
        params = [p for m in lin for p in (m.weight, m.bias)]
        key = tuple((p.data_ptr(), p._version) for p in params)
        cached = getattr(self, "_pack_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1], cached[2]
        n = len(lin)
        widths = (C.c_int32 * (n + 1))(lin[0].in_features, *[m.out_features for m in lin])
        nbytes = _lib.lib().sdirt_mlp_packed_bytes(widths, n)
        if nbytes < 0:
            raise _lib.SdirtError(_lib.lib().sdirt_last_error().decode())
        dev = lin[0].weight.device
        buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ws = [m.weight.detach().float().contiguous() for m in lin]
        bs = [m.bias.detach().float().contiguous() for m in lin]
        _lib.check(_lib.lib().sdirt_mlp_pack((C.c_void_p * n)(*[w.data_ptr() for w in ws]),
                                             (C.c_void_p * n)(*[b.data_ptr() for b in bs]),
                                             widths, n, dptr(buf), stream_ptr(dev)))
        self._pack_cache = (key, buf, widths)
        return buf, widths

    @torch.no_grad()
    def forward_fused(self, inp, mirror=False):
        """MLP.forward under fp16 autocast, all layers in one kernel with the activations held in
        LDS (csrc/sdirt_mlp.hip).  inp [..., 3] fp32 on the GPU -> fp16 [..., ks, ks]; with
        mirror=True -> [2, ..., ks, ks]: the network at (x, y, z) and at (-x, y, z), the two
        passes of PSFNet.pred."""
        if not inp.is_cuda:
            raise _lib.SdirtError("forward_fused runs on the GPU only")
        buf, widths = self._packed()
        flat = inp.detach().reshape(-1, 3).to(torch.float32).contiguous()
        n, of = flat.shape[0], int(widths[len(widths) - 1])
        out = torch.empty(((2 if mirror else 1) * n, of), dtype=torch.float16, device=inp.device)
        _lib.check(_lib.lib().sdirt_psfnet_mlp(dptr(buf), widths, len(widths) - 1, dptr(flat), n,
                                               1 if mirror else 0, dptr(out), stream_ptr(inp.device)))
        lead = tuple(inp.shape[:-1])
        tail = (self.ks, self.ks) if self.ks * self.ks == of else (of,)
        return out.reshape(((2,) if mirror else ()) + lead + tail)


class _BilinearUp(nn.Module):
    def __init__(self, scale_factor):
        super().__init__()
        self.scale_factor = scale_factor

    def forward(self, x):
        return nnF.interpolate(x, scale_factor=self.scale_factor, mode="bilinear",
                               align_corners=False)


class MLPConv(nn.Module):
    def __init__(self, in_features, ks, activation="relu", channels=1):
        super().__init__()
        self.ks, self.ks_mlp, self.channels = ks, ks // 4, channels
        enc = [in_features, 256, 256, 512]
        mods = []
        for a, b in zip(enc[:-1], enc[1:]):
            mods += [nn.Linear(a, b), nn.ReLU()]
        mods.append(nn.Linear(512, channels * self.ks_mlp ** 2))
        self.encoder = nn.Sequential(*mods)

        def up(cin, cout):
            return nn.ConvTranspose2d(cin, cout, kernel_size=3, stride=1, padding=1)
        self.decoder = nn.Sequential(
            up(channels, 64), nn.ReLU(), up(64, 64), nn.ReLU(), nn.Upsample(scale_factor=2),
            up(64, 64), nn.ReLU(), up(64, 64), nn.ReLU(), _BilinearUp((2.1, 2.1)),
            up(64, 64), nn.ReLU(), up(64, channels))
        self.activation = {"relu": nn.ReLU, "sigmoid": nn.Sigmoid}[activation]()

    def forward(self, x):
        seed = self.encoder(x).view(-1, self.channels, self.ks_mlp, self.ks_mlp)
        out = self.activation(self.decoder(seed))[:, 0]
        return out.view(*x.shape[:-1], out.shape[-2], out.shape[-1])
