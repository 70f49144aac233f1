"""Every committed fixture must regenerate from the committed recipe.

Runs ``tests/golden/make_golden.py``, ``make_golden_fdr.py`` and ``make_golden_chain.py`` (the reference itself, imported through
``ref_shim``) into a temporary directory and compares every array with ``tests/golden/*.npz``.  Build
container only: the reference tree does not exist on the GPU box (the test skips there).
"""

from __future__ import annotations

import glob
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
REFERENCE = "/root/reference"


def _same(a: np.ndarray, b: np.ndarray) -> bool:
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype.kind in "fc":
        return bool(np.array_equal(a, b, equal_nan=True))
    return bool(np.array_equal(a, b))


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "alphadia")), reason="reference tree not present")
def test_fixtures_regenerate_from_the_committed_recipe(tmp_path):
    env = dict(os.environ)
    env.pop("ADH_LIB_PATH", None)
    for script in ("make_golden.py", "make_golden_fdr.py", "make_golden_chain.py"):
        r = subprocess.run([sys.executable, os.path.join(GOLDEN, script), "--out", str(tmp_path)],
                           capture_output=True, text=True, env=env, timeout=1200)
        assert r.returncode == 0, r.stderr[-2000:]
    committed = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "*.npz")))
    regenerated = sorted(os.path.basename(p) for p in glob.glob(os.path.join(str(tmp_path), "*.npz")))
    assert committed == regenerated
    different = []
    for name in committed:
        old = np.load(os.path.join(GOLDEN, name), allow_pickle=True)
        new = np.load(os.path.join(str(tmp_path), name), allow_pickle=True)
        if sorted(old.files) != sorted(new.files):
            different.append(f"{name}: keys differ")
            continue
        for key in old.files:
            a, b = old[key], new[key]
            if key == "caveat":
                continue  # (names the NumPy version the fixture was made with)
            if key.endswith("_columns") and a.dtype.kind in "OUS":
                if sorted(map(str, a.tolist())) != sorted(map(str, b.tolist())):  # the tests compare them sorted
                    different.append(f"{name}:{key}")
                continue
            if not _same(a, b):
                different.append(f"{name}:{key}")
    assert not different, "fixtures that do not regenerate: " + ", ".join(different[:20])
