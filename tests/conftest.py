import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(GOLDEN, "kat.json"), encoding="utf-8") as f:
        return json.load(f)


@pytest.fixture(scope="session")
def sv_embed():
    return np.load(os.path.join(GOLDEN, "sensevoice_embed.npy"))


def _nan(x):
    return [float("nan") if v == "nan" else float(v) for v in x]


@pytest.fixture(scope="session")
def nanlist():
    return _nan
