import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def images():
    z = np.load(os.path.join(ROOT, "tests", "golden", "images.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    z = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def orbx():
    """The product package with liborbx.so built (CPU-side checks may load it without a GPU)."""
    try:
        import torch  # noqa: F401  -- load order: torch bundles its own libamdhip64; whichever copy is loaded first serves the
        # whole process, and torch does not find the GPU on the system's copy
    except ImportError:
        pass
    import orb_slam_tracking_amd as pkg
    if not os.path.exists(pkg.lib_path()):
        import __graft_entry__
        __graft_entry__.build()
    pkg.lib()
    return pkg
