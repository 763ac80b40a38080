import json
import os
import sys

import numpy as np
import pytest
import torch

TESTS = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(TESTS)
GOLDEN = os.path.join(TESTS, "golden")
PKG = os.path.join(ROOT, "multichannel-semseg-with-uda_amd")

# The product mirrors the reference's flat script layout (``import loss``, ``from models.model_util
# import get_models`` ...), so its directory goes first on sys.path -- ahead of site-packages, which
# holds an unrelated HuggingFace ``datasets``.
for p in (GOLDEN, ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
for name in ("datasets", "loss", "util", "models"):
    mod = sys.modules.get(name)
    if mod is not None and not getattr(mod, "__file__", "").startswith(PKG):
        del sys.modules[name]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def has_gpu():
    return torch.cuda.is_available()


@pytest.fixture(scope="session")
def golden():
    class G:
        dir = GOLDEN

        @staticmethod
        def npz(name):
            return np.load(os.path.join(GOLDEN, name))

        @staticmethod
        def json(name):
            with open(os.path.join(GOLDEN, name)) as fh:
                return json.load(fh)
    return G


@pytest.fixture(scope="session")
def device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
