import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)

GOLDEN = os.path.join(TESTS, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): compile libkvfe.so for gfx950 once
    (hipcc cross-compiles without a GPU; ~30 s) so that the suite does not depend on build() having run first.
    Building is not a fallback: without the library every product call still fails loudly."""
    so = os.path.join(ROOT, "kimera_vio_amd", "csrc", "libkvfe.so")
    if not os.path.exists(so):
        import subprocess
        subprocess.run(["make", "-C", os.path.dirname(so), "-j8"], check=True, capture_output=True)


@pytest.fixture(scope="session")
def golden():
    return GOLDEN
