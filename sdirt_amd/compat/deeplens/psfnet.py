"""deeplens.psfnet -> sdirt_amd.psfnet."""
from sdirt_amd.psfnet import PSFNet  # noqa: F401
from .optics import *           # noqa: F401,F403
from .render_psf import *       # noqa: F401,F403
from .psfnet_arch import *      # noqa: F401,F403

DMIN = 200     # [mm], psfnet.py:15
DMAX = 20000   # [mm], psfnet.py:16
