import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """A fresh checkout has no built libraries (they are git-ignored): build what the suites load -- the HIP library
    (hipcc cross-compiles without a GPU), the host-side ingest library and the C oracle -- once, if stale or missing."""
    import warnings
    try:
        from invpref_kdd_2022_amd import build
        build.build()
    except Exception as exc:  # the tests that load the library then fail on their own, with the loader's message
        warnings.warn(f'could not build the HIP / ingest libraries: {exc}')
    try:
        from oracle import oracle
        oracle.build()
    except Exception as exc:
        warnings.warn(f'could not build the CPU oracle: {exc}')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
