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
for p in (TESTS, GOLDEN, ROOT, PKG):
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


LIB_OPTIONS = ("BN_STATS_ONE", "BN_REVERSE", "BN_V4", "BIGTILE_MIN_SLOTS", "WIDETILE_MIN_SLOTS", "DGRAD_INTERLEAVE", "DGRAD_ADD_LDS", "PACK_BLOCKS",
               "PP_MIN_ROUNDS", "PP_CUS", "PINGPONG", "PP_WIDE_FILL", "PP_WIDE128", "PP_DEEP", "THIN_WINDOW", "WGRAD_WGS", "WGRAD_THIN_TR", "WGRAD_TR64",
               "WGRAD_TR", "WGRAD_BIG", "WGRAD_TWOTAP", "WGRAD_PP_CUS", "WGRAD_PP", "WGRAD_PP3", "WGRAD_PP_DEEP", "UP8_LOSS_DMA", "UP8_BAND_ROWS")


@pytest.fixture
def libopt():
    """``libopt(PINGPONG=0, PP_CUS=16)``: library options through the ABI (mcdseg_set_option -- the library reads no environment
    variable); whatever a test set is restored when it ends"""
    import mcdseg
    first = {}

    def set_(**values):
        for k, v in values.items():
            prev = mcdseg.set_option(k, int(v))
            first.setdefault(k, prev)
    yield set_
    for k, v in first.items():
        mcdseg.set_option(k, v)
