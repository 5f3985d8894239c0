import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


# the test suite checks the product library only: a PO_LIB_PATH left over from tools/exp/ab.sh is dropped (and said so)
for _v in ("PO_LIB_PATH", "PO_ALLOW_VARIANT"):
    if os.environ.pop(_v, None) is not None:
        sys.stderr.write("tests/conftest.py: ignoring %s from the environment\n" % _v)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
