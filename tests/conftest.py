import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.binding import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The reference's own object code (oracle/_ref). Container only."""
    from oracle.binding import Ref, ref_available

    if not ref_available():
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return Ref()


@pytest.fixture()
def ctx():
    """A FRESH device context through the C-ABI for every test (statistics hint, tier steering and XCD balance start from
    scratch: which kernel variant a test reaches does not depend on what ran before it).  Fails loudly if the HIP library
    is missing."""
    import hdr2yuv_amd as h

    c = h.Context(0)
    yield c
    c.close()
