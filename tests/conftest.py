import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the native pieces live in the tree (git-ignored); build whatever a fresh checkout lacks
    need = [os.path.join(ROOT, "oswald_amd", f) for f in ("liboswald_hip.so", "liboswald_host.so", "oswald", "liboswald_hip.isa.json")]
    need.append(os.path.join(ROOT, "oracle", "liboswald_oracle.so"))
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()
    else:
        # a library rebuilt by hand (make) after the last ISA check: check and stamp it again (tools/isa_check.py)
        import hashlib
        import json
        stamp = json.load(open(need[3]))
        if stamp.get("library_sha256") != hashlib.sha256(open(need[0], "rb").read()).hexdigest():
            # ... where there is a compiler; a box that received a prebuilt library without one leaves the mismatch to
            # tests/test_gpu_isa_guard.py, which reports it as a test failure instead of aborting the session here
            import shutil
            import subprocess
            if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
                subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "isa_check.py")], stdout=subprocess.DEVNULL)


def pytest_generate_tests(metafunc):
    # every GPU test runs once per first-pass arithmetic of the library; a test that passes cell_bits explicitly
    # is unaffected by the setting
    if metafunc.definition.get_closest_marker("gpu") and "first_pass" in metafunc.fixturenames:
        metafunc.parametrize("first_pass", ["i16", "i16plain"], indirect=True)


@pytest.fixture(autouse=True)
def first_pass(request, monkeypatch):
    mode = getattr(request, "param", None)
    if mode is not None:
        # i16: the default (column-frame int16 cell with the plain biased cell as fallback);
        # i16plain: the plain biased int16 cell only
        monkeypatch.setenv("OSWALD_HIP_CELL_BITS", "16")
        if mode == "i16plain":
            monkeypatch.setenv("OSWALD_HIP_NO_FRAME", "1")
        else:
            monkeypatch.delenv("OSWALD_HIP_NO_FRAME", raising=False)
    return mode


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    pyoracle.load()
    return pyoracle


@pytest.fixture(scope="session")
def toppin(oracle):
    """oracle/toppin.py: a top list pinned to the oracle's scalar scores (the metric's "top-10 score bit-exact")."""
    import toppin as tp
    return tp


def _two_device_ids(kind):
    """Device ids of a two-device context: "aliased" = the one GPU of the test box twice (per-device streams, buffers
    and queues, level-1 gather on one GPU); "distinct" = GPUs 0 and 1 -- different devices, peer traffic and the RCCL
    all-gather of oswald_hip_topr -- on a box that has them."""
    if kind == "aliased":
        return [0, 0]
    from oswald_amd import capi
    if capi.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (this box has %d): runs by itself on a multi-GPU node" % capi.device_count())
    return [0, 1]


@pytest.fixture(params=["aliased", "distinct"])
def two_devices(request):
    return _two_device_ids(request.param)


@pytest.fixture(scope="session")
def hip_ctx():
    """One C-ABI context for the whole GPU session (the library is loaded from
    the tree; a missing library or GPU is a hard failure, not a skip)."""
    from oswald_amd import capi
    ctx = capi.Context(1)
    yield ctx
    ctx.close()
