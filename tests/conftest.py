import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gst-plugins-rs_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/liboracle.so), built on demand. Checker only."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def synth():
    from mi355fx import synth as S
    return S


@pytest.fixture(scope="session")
def mi355lib():
    """libmi355fx.so must already be built (make -C gst-plugins-rs_amd); build it if hipcc is here."""
    import subprocess
    import mi355fx
    if not os.path.exists(mi355fx.LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", PKG])
    return mi355fx.load_library()


@pytest.fixture()
def ctx(mi355lib):
    """A device context. On a GPU box this MUST succeed: no skipping, no fallback."""
    import mi355fx
    c = mi355fx.Context(0)
    yield c
    c.close()
