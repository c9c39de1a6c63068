import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name))
    out = {}
    for k in z.files:
        a = z[k]
        t = torch.from_numpy(a)
        out[k] = t
    return out


@pytest.fixture(scope="session")
def scene():
    return load_npz("scene.npz")


@pytest.fixture(scope="session")
def weights():
    return load_npz("weights.npz")


@pytest.fixture(scope="session")
def golden_fpn():
    return load_npz("fpn.npz")


@pytest.fixture(scope="session")
def golden_pipe():
    return load_npz("pipeline.npz")


@pytest.fixture(scope="session")
def golden_render():
    return load_npz("render.npz")


@pytest.fixture(scope="session")
def golden_perturb():
    return load_npz("render_perturb.npz")


@pytest.fixture(scope="session")
def golden_validate():
    """ImplicitSurface.validate's image outputs from the reference itself (tests/golden/make_golden.py)."""
    return load_npz("validate.npz")


@pytest.fixture(scope="session")
def golden_grid():
    return load_npz("sdf_grid.npz")


@pytest.fixture(scope="session")
def golden_train():
    return load_npz("train_outputs.npz")


@pytest.fixture(scope="session")
def golden_grads():
    """The reference's own loss.backward() through its own render (tests/golden/make_golden_grad.py)."""
    return load_npz("train_grads.npz")


@pytest.fixture(scope="session")
def golden_vgrads():
    """The reference's autograd through its own FPN / cost volume / sparse2dense / matching field / compute_ptloss."""
    return load_npz("volume_grads.npz")
