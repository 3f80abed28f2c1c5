import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_artifacts():
    """The .so files are git-ignored: build them (hipcc cross-compiles without a GPU) if a fresh checkout lacks any."""
    need = [os.path.join(ROOT, "altair-raytracing_amd", "csrc", "libisx.so"),
            os.path.join(ROOT, "altair-raytracing_amd", "host", "isx_macro"),
            os.path.join(ROOT, "oracle", "libisx_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_fixtures.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def isx():
    """The product library bound to cuda:0 / HIP device 0.  Fails (not skips) if it cannot load."""
    import altair_raytracing_amd as m
    m.load()
    m.init(0)
    yield m
    m.shutdown()
