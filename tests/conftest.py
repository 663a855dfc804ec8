import glob
import os
import subprocess
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


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def golden_names(prefix=""):
    return sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLDEN, prefix + "*.npz"))
                  if "edge" not in f and "kernels_" not in f and "oracle_" not in os.path.basename(f))


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); builds its C part on first use."""
    lib = os.path.join(ROOT, "oracle", "build", "librbf_oracle.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    import gp_oracle
    return gp_oracle


@pytest.fixture(scope="session")
def ctx():
    """One GPU context shared by the gpu tests; loading fails loudly without the .so."""
    from gaussian_process_amd import GPContext
    c = GPContext(0)
    yield c
    c.close()
