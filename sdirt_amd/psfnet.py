"""The PSF-producing half of the reference's PSFNet (deeplens/psfnet.py): the
lens state it sets up and the training / test data generators that call the
ray-traced PSF.  The MLP itself, its optimiser loop, checkpoints and the image
renderer built on it are consumers of this path (stock PyTorch) and are not
re-implemented here; `get_training_data` hands them batches straight from the
HIP kernels, already on the GPU.
"""
import numpy as np
import torch

from .optics import Lensgroup

DMIN = 200      # [mm]  psfnet.py:15
DMAX = 20000    # [mm]  psfnet.py:16

_PSFNET_SENSOR_Z = {"rf35mm": 80.447, "rf50mm": 62.25}          # psfnet.py:42-48


class PSFNet(Lensgroup):
    """psfnet.py:18-57 without the network: Lensgroup + kernel size + depth range."""

    def __init__(self, filename, model_name="mlp", kernel_size=11, sensor_res=(512, 512),
                 device="cuda"):
        super().__init__(filename=filename, sensor_res=sensor_res, device=device)
        self.kernel_size = kernel_size
        self.model_name = model_name
        self.d_max = -DMAX
        self.d_min = -DMIN
        for key, z in _PSFNET_SENSOR_Z.items():
            if filename.find(key) != -1:
                self.d_sensor = z
                break
        else:
            raise ValueError("filename is not correct")          # psfnet.py:46-48 prints and exits
        self.foc_d_arr = np.array([-999.9, -1000, -1000.1], dtype=np.float32) + self.d_sensor
        self.foc_z_arr = (self.foc_d_arr - self.d_min) / (self.d_max - self.d_min)
        self.foc_d = np.array([-1000.0], dtype=np.float32) + self.d_sensor

    def depth2z(self, depth):
        z = (depth - self.d_min) / (self.d_max - self.d_min)
        return torch.clamp(z, min=0, max=1)

    def z2depth(self, z):
        return z * (self.d_max - self.d_min) + self.d_min

    def _warp_z(self, z_gauss, foc_z):
        # psfnet.py:190-193 / 229-232
        z = torch.zeros_like(z_gauss)
        z[z_gauss > 0] = (1 - foc_z) * z_gauss[z_gauss > 0] / 3 + foc_z
        z[z_gauss < 0] = foc_z * z_gauss[z_gauss < 0] / 3 + foc_z
        return z

    def get_training_data(self, bs=256, spp=4096):
        """psfnet.py:170-202: (inp [bs,3] in [-1,1]^2 x [0,1], psf [bs,ks,ks] on the GPU)."""
        foc_z = np.random.choice(self.foc_z_arr)
        x = (torch.rand(bs) - 0.5) * 2
        y = (torch.rand(bs) - 0.5) * 2
        z = self._warp_z(torch.clamp(torch.randn(bs), min=-3, max=3), foc_z)
        inp = torch.stack((x, y, z), dim=-1)
        points = torch.stack((x, y, self.z2depth(z)), dim=-1)
        return inp, self.psf(points=points, ks=self.kernel_size, spp=spp)

    def get_test_data(self, bs=1024, spp=65536):
        """psfnet.py:204-241: 32x32 grid paired ELEMENT-WISE with bs depths."""
        foc_z = self.foc_z_arr[1]
        g = 32
        x, y = torch.meshgrid(torch.linspace(-1 + 1 / (2 * g), 1 - 1 / (2 * g), g),
                              torch.linspace(1 - 1 / (2 * g), -1 + 1 / (2 * g), g), indexing="xy")
        x, y = x.reshape(-1), y.reshape(-1)
        z = self._warp_z(torch.linspace(-3, 3, bs), foc_z)
        inp = torch.stack((x, y, z), dim=-1)
        points = torch.stack((x, y, self.z2depth(z)), dim=-1)
        return inp, self.psf(points=points, ks=self.kernel_size, spp=spp)
