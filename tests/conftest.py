import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(ROOT, "scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def scene_path(name):
    return os.path.join(SCENES, name)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    orc.set_threads(min(8, os.cpu_count() or 1))
    return orc


def rel_err(a, b, floor=1e-12):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)
