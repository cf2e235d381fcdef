"""The pin of the oracle is the reference itself, compiled in the build container (oracle/Makefile `ref`) and run by
oracle/gen_golden.py.  Wherever the reference's sources are present this test does that again -- into a scratch directory
-- and compares every fixture with the committed one: tests/golden/ is what the reference computes TODAY, not what it once
did.  (On the GPU box there is no /root/reference and nothing of it is loaded: skipped.)"""
import filecmp
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/host/src"
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.mark.skipif(not os.path.isdir(REF_SRC), reason="the reference's sources are not on this machine")
def test_golden_fixtures_regenerate_from_the_compiled_reference(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    assert os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libosw_ref.so"))
    out = tmp_path / "golden"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py"), str(out)], stdout=subprocess.DEVNULL, timeout=600)
    committed = sorted(f for f in os.listdir(GOLDEN) if f.endswith((".json", ".npz")))
    assert committed == sorted(os.listdir(out)) and len(committed) == 8
    for f in committed:
        a, b = os.path.join(GOLDEN, f), str(out / f)
        if filecmp.cmp(a, b, shallow=False):
            continue
        # (a zip container may differ in its metadata between library versions: then the arrays must still be the same)
        assert f.endswith(".npz"), f"{f} differs from what the compiled reference produces"
        with np.load(a) as x, np.load(b) as y:
            assert sorted(x.files) == sorted(y.files), f
            for k in x.files:
                np.testing.assert_array_equal(x[k], y[k], err_msg=f"{f}:{k}")
