"""Parity gate 1 (GPU): the HIP path, called through the C ABI, must reproduce every reference golden image of
the hot path with tolerance 0 — the same bar the reference's own assert_golden applies to its CPU path
(reference tests/common/mod.rs:181-186).  This holds even for the ±1 LSB class (Gaussian with fused multiply-add)
on these inputs; the class tolerance itself is exercised in test_gpu_parity.py."""
import numpy as np
import pytest

from . import golden_cases as GC

pytestmark = pytest.mark.gpu
CASES = GC.all_cases()


@pytest.fixture(scope="module")
def gpu():
    from .backends import GpuBackend
    return GpuBackend(0)


@pytest.mark.parametrize("key,method,kwargs", CASES, ids=[c[0] for c in CASES])
def test_hip_matches_reference_golden(gpu, golden, key, method, kwargs):
    out = getattr(gpu, method)(**kwargs)
    exp = golden[key]
    assert out.shape == exp.shape
    diff = np.abs(out.astype(np.int16) - exp.astype(np.int16))
    assert diff.max() == 0, f"{key}: {int((diff.max(-1) > 0).sum())} px differ, max channel diff {int(diff.max())}"
