import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The tests assert on hdrnet_last_kernel(); its bookkeeping is off by default (include/hdrnet_amd.h).
os.environ.setdefault("HDRNET_AMD_KERNEL_NAMES", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # A `gpu` test on a box without a GPU is an error in how the suite was invoked,
    # not something to skip silently -- except in the CPU container, where the driver
    # deselects them with -m "not gpu" anyway.
    pass


@pytest.fixture(scope="session")
def port():
    import oracle
    return oracle.port()


@pytest.fixture(scope="session")
def ref_or_none():
    import oracle
    try:
        return oracle.ref()
    except FileNotFoundError:
        return None


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz")
                  and "kat" not in f)
