"""deeplens.optics -> sdirt_amd.optics (the star import of the reference also hands on numpy, torch and the
constants of deeplens.basics, which its scripts use unqualified)."""
import numpy as np              # noqa: F401
import torch                    # noqa: F401
import torch.nn as nn           # noqa: F401
from sdirt_amd.optics import Lensgroup  # noqa: F401
from .basics import *           # noqa: F401,F403
