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


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """The CPU oracle's small convolutions run slower on ALL hardware threads of a many-core host than on a few (bench.py's
    cpu_baseline measures the same): cap torch's intra-op threads for the suite."""
    import torch
    n = torch.get_num_threads()
    torch.set_num_threads(max(1, min(n, 16)))
    yield
    torch.set_num_threads(n)


@pytest.fixture(scope="session")
def oracle_fullsize_nf64():
    """BASELINE configs[1]'s utterance through the CPU oracle once per session: 4 s / 8 kHz / 2 speakers, nf = 64, N = 30 + 1
    corrector step = 60 network evaluations on injected noise (synthetic weights seed 7).  The fp32, f16 and split engines are
    all gated against this one result (tests/test_engine_gpu.py, tests/test_split_gpu.py)."""
    import torch
    import diffsep_oracle as O
    from diffsep_amd import synth
    torch.set_grad_enabled(False)
    cfg = O.default_config(64, 2)
    T, B, N = 32000, 1, 30
    sd = synth.synth_state_dict(O.param_table(cfg), 7)
    mix = torch.from_numpy(synth.synth_batch(B, T=T)[0])
    draws = [torch.from_numpy(synth.synth_noise(f"fs.z{i}", (B, 2, T))) for i in range(1 + 2 * N)]
    ref, nfe = O.separate(O.to_torch(sd), cfg, mix, draws, N=N, corrector_steps=1, snr=0.5, eps=0.03, denoise=True)
    return dict(mix=mix, draws=draws, ref=ref, nfe=nfe, N=N, T=T, B=B)
