import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "diffusion-separation_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json

    import numpy as np
    data = dict(np.load(os.path.join(GOLDEN_DIR, "golden_ref.npz")))
    with open(os.path.join(GOLDEN_DIR, "golden_meta.json")) as f:
        meta = json.load(f)
    return data, meta


@pytest.fixture(scope="session")
def golden2():
    """round-2 vectors (tests/golden/gen_golden_r2.py): SDE surface, nf = 128, three sources"""
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN_DIR, "golden_ref2.npz")))


@pytest.fixture(scope="session")
def golden3():
    """round-4 vectors (tests/golden/gen_golden_r4.py): the time embedding in isolation"""
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN_DIR, "golden_ref3.npz")))
