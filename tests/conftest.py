"""pytest configuration: the ``gpu`` marker, import paths, shared helpers.

* ``-m "not gpu"``: oracle vs golden fixtures, host logic, C-ABI symbol checks, gloo world-size-2.
* ``-m gpu``: parity tests proper -- the HIP path (through the C-ABI) vs oracle / golden fixtures.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "so-net_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def pytest_collection_modifyitems(config, items):
    """GPU tests must never silently pass on a box without a GPU: skip them there, loudly."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (run with gpurun)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def assert_close_rms(got, ref, rel=1e-5, what=""):
    """|got-ref| <= rel * max(|ref|, rms(ref))  -- the float metric of SURVEY.md 7 (hard part 4)."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    rms = float(np.sqrt(np.mean(ref ** 2))) if ref.size else 0.0
    bound = rel * np.maximum(np.abs(ref), rms)
    err = np.abs(got - ref)
    bad = err > bound
    assert not bad.any(), "%s: %d/%d elements off, worst err/bound = %.3g (rms %.3g)" % (
        what, int(bad.sum()), bad.size, float((err / np.maximum(bound, 1e-300)).max()), rms)
