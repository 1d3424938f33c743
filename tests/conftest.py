import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


os.environ.setdefault('FK_BACKTRACE', '1')      # a crash inside libfawkes_hip.so prints its native backtrace (read by fk_init)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def ctx():
    """The HIP context.  No fallback: on a GPU box a missing library or device is a test failure."""
    import fawkes_crypto_amd as fk
    c = fk.Context(0)
    yield c
    c.close()


@pytest.fixture(scope='session')
def oracle():
    import c_oracle
    c_oracle.build()
    c_oracle.lib()
    return c_oracle
