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


class Fixture:
    """A golden .npz: arrays grouped by the prefix before '/', plus JSON meta."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.meta = json.loads(bytes(z["__meta__"]).decode())
        self.groups = {}
        for k in z.files:
            if k == "__meta__":
                continue
            g, _, rest = k.partition("/")
            if not rest:
                g, rest = "", k
            self.groups.setdefault(g, {})[rest] = torch.from_numpy(np.array(z[k]))

    def __getitem__(self, g):
        return self.groups.get(g, {})


def load_fixture(name):
    return Fixture(name)


def rel_l2(a, b):
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="session")
def golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))
