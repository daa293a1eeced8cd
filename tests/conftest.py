import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "sdirt_amd", "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def load_state(lens):
    with open(os.path.join(GOLDEN, f"lens_state_{lens}.json")) as f:
        return json.load(f)


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc


def make_lens(name, device, state=None):
    """Lensgroup on the package's own prescription file with the geometric-optics
    scalars pinned to the reference fixture (lens_state_*.json)."""
    from sdirt_amd import Lensgroup
    st = state or load_state(name)
    path = os.path.join(DATA, f"{name}.json")
    if not os.path.exists(path):                # test-only prescriptions live with the fixtures
        path = os.path.join(GOLDEN, f"lens_{name}.json")
    lens = Lensgroup(path, sensor_res=(512, 768), post_computation=False, device=device)
    lens.set_state(d_sensor=st["d_sensor"], hfov=st["hfov"],
                   pupil=(st["pupil_z"], st["pupil_r"]),
                   exit_pupil=(st["exit_pupil_z"], st["exit_pupil_r"]))
    return lens
