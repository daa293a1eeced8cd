"""Run-script helpers with the reference's names (deeplens/utils.py:136-164)."""
import logging
import os
import random

import numpy as np
import torch


def set_seed(seed=0):
    """utils.py:136-145: seed python, numpy and torch (CPU generator -- the one the pupil samples are drawn
    from -- and the current GPU's)."""
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def set_logger(dir="./"):
    """utils.py:148-164: INFO records to the console and to <dir>/output.log."""
    root = logging.getLogger()
    root.setLevel("DEBUG")
    fmt = logging.Formatter("%(asctime)s:%(levelname)s:%(message)s", "%Y-%m-%d %H:%M:%S")
    for handler in (logging.StreamHandler(), logging.FileHandler(os.path.join(dir, "output.log"))):
        handler.setFormatter(fmt)
        handler.setLevel("INFO")
        root.addHandler(handler)
