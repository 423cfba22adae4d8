import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", params=["mfma", "valu"])
def engine(request):
    """The HIP engine, once per Hamming backend (fp4 Gram matrix on the matrix cores / XOR + popcount on the
    VALU: both exact).  No fallback: on a box without a GPU (or without libvdf_hip.so) this raises."""
    import vid_dup_finder_lib_amd as vdf

    old = os.environ.get("VDF_SEARCH_BACKEND")
    os.environ["VDF_SEARCH_BACKEND"] = request.param
    try:
        eng = vdf.Engine(0)
    finally:
        if old is None:
            os.environ.pop("VDF_SEARCH_BACKEND", None)
        else:
            os.environ["VDF_SEARCH_BACKEND"] = old
    eng.backend = request.param
    yield eng
    eng.close()
