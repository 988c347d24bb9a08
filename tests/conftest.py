import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def golden_ml100k():
    return load_golden("ml100k_known_answer.npz")


@pytest.fixture(scope="session")
def golden_eval():
    return load_golden("eval_cases.npz")


@pytest.fixture(scope="session")
def golden_sampler():
    return load_golden("sampler.npz")


@pytest.fixture(scope="session")
def golden_steps():
    return load_golden("train_steps.npz")


def csr_from_pairs(u, i, U, I):
    import scipy.sparse as sp
    m = sp.csr_matrix((np.ones(len(u), np.float32), (u.astype(np.int64), i.astype(np.int64))), shape=(U, I))
    m.sort_indices()
    return m
