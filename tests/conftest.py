import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Flat npz -> {case: {key: array}}."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
        self.cases = {}
        for k in z.files:
            case, _, key = k.partition("/")
            self.cases.setdefault(case, {})[key] = z[k]

    def __getitem__(self, case):
        return self.cases[case]

    def names(self):
        return sorted(self.cases)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return load


def parse_pairs(arr):
    out = []
    for s in arr.tolist():
        a, b, w = s.split("|")
        out.append(((a, b), float(w)))
    return out


def parse_flags(arr):
    out = {}
    for s in arr.tolist():
        k, v = s.split("=")
        out[k] = v == "True"
    return out
