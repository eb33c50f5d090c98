import json
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


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def manifest(npz, key="manifest"):
    return {k: tuple(v) for k, v in json.loads(bytes(npz[key]).decode()).items()}


def rel_l2(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def load_synth(module, seed=0):
    """Fill a module with the deterministic synthetic weights used by the goldens."""
    from ln3diff_amd.synth import load_synth_
    return load_synth_(module, seed)


@pytest.fixture(scope="session")
def hip_lib():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build()
    from ln3diff_amd import _lib
    return _lib.lib()


@pytest.fixture(scope="session")
def hip_lib_path():
    """Path of the built in-tree library (build() cross-compiles without a GPU); nothing is dlopen'ed for compute."""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build()
    return ge.LIB
