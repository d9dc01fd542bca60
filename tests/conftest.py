import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs oracle/_ref/libt3ref.so (the reference compiled in place)")


@pytest.fixture(scope="session")
def orc():
    import oracle_lib
    oracle_lib.build_oracle()
    return oracle_lib.oracle()


@pytest.fixture(scope="session")
def ref(orc):
    import oracle_lib
    if not oracle_lib.have_ref():
        pytest.skip("oracle/_ref/libt3ref.so not built (no /root/reference here)")
    return oracle_lib.Ref()


@pytest.fixture(scope="session")
def t3():
    """The product's host-side mirror (ternary-image-codec_amd), loaded through __graft_entry__."""
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def gpu(t3):
    """Initialised device context; fails loudly (no fallback) when the HIP extension or device is missing."""
    t3.init(0)
    return t3
