"""Import the reference (LinYark/Sdirt) from /root/reference with stub modules.

TEST INFRASTRUCTURE ONLY.  Works only in the build container (the reference is
not present on the GPU box).  Used by oracle/gen_golden.py to produce the
committed fixtures under tests/golden/.  Recipe: SURVEY.md Appendix B.
"""
import sys
import types

import torch

REF_ROOT = "/root/reference"


def _m(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _stub(*a, **k):
    raise RuntimeError("stubbed: not on the PSF path")


def import_reference(num_threads=1):
    """Returns (PSFNet class, set_seed, deeplens module)."""
    torch.set_num_threads(num_threads)
    if "deeplens" in sys.modules:
        import deeplens
        from deeplens.psfnet import PSFNet
        from deeplens.utils import set_seed
        return PSFNet, set_seed, deeplens
    _m("cv2"); _m("lpips"); _m("imageio")
    sk = _m("skimage")
    sk.metrics = _m("skimage.metrics", peak_signal_noise_ratio=_stub,
                    structural_similarity=_stub)
    tv = _m("torchvision")
    tv.utils = _m("torchvision.utils", save_image=_stub, make_grid=_stub)
    tv.transforms = _m("torchvision.transforms")
    tv.transforms.functional = _m("torchvision.transforms.functional")
    sys.path.insert(0, REF_ROOT)
    sys.dont_write_bytecode = True
    import deeplens
    from deeplens.psfnet import PSFNet
    from deeplens.utils import set_seed
    return PSFNet, set_seed, deeplens
