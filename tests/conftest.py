import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gold():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLD, name), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def gold_json():
    import json

    def load(name):
        with open(os.path.join(GOLD, name)) as f:
            return json.load(f)
    return load


def ctc_case_names(npz):
    return sorted({k.split("/")[0] for k in npz.files})
