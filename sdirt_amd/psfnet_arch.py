"""Network architectures that represent the PSF field (consumers of the ray-traced PSFs).

Parameter names and shapes follow deeplens/psfnet_arch.py so that the reference's
checkpoints (`F4_PSFNet_mlp.pkl`: keys `net.<2i>.weight/bias`) load unchanged:
  MLP      psfnet_arch.py:26-50   in -> h/4 -> h -> (h -> h) x L -> out, ReLU after every layer
  MLPConv  psfnet_arch.py:76-136  MLP encoder to a (ks//4)^2 seed, transposed-conv decoder
The layers are stock torch.nn (hipBLASLt / MIOpen GEMMs on ROCm); the fp16 autocast the
reference wraps MLP.forward in (psfnet_arch.py:46) is applied on CUDA devices only.
"""
import torch
import torch.nn as nn
import torch.nn.functional as nnF


def initialize_weights(m):
    """psfnet_arch.py:291-303."""
    if isinstance(m, nn.Conv2d):
        nn.init.kaiming_uniform_(m.weight.data, nonlinearity="relu")
        if m.bias is not None:
            nn.init.constant_(m.bias.data, 0)
    elif isinstance(m, nn.BatchNorm2d):
        nn.init.constant_(m.weight.data, 1)
        nn.init.constant_(m.bias.data, 0)
    elif isinstance(m, nn.Linear):
        nn.init.kaiming_uniform_(m.weight.data)
        nn.init.constant_(m.bias.data, 0)
    elif isinstance(m, nn.ConvTranspose2d):
        nn.init.xavier_uniform_(m.weight)


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
