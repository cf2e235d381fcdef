import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the native pieces live in the tree (git-ignored); build whatever a fresh checkout lacks
    need = [os.path.join(ROOT, "oswald_amd", f) for f in ("liboswald_hip.so", "liboswald_host.so", "oswald")]
    need.append(os.path.join(ROOT, "oracle", "liboswald_oracle.so"))
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    pyoracle.load()
    return pyoracle


@pytest.fixture(scope="session")
def hip_ctx():
    """One C-ABI context for the whole GPU session (the library is loaded from
    the tree; a missing library or GPU is a hard failure, not a skip)."""
    from oswald_amd import capi
    ctx = capi.Context(1)
    yield ctx
    ctx.close()
