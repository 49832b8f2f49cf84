"""Pin the CPU oracle: every known-answer case must equal the reference's golden image with tolerance 0
(reference tests/common/mod.rs:181-186: GOLDEN_TOLERANCE defaults to 0)."""
import numpy as np
import pytest

from . import golden_cases as GC
from .backends import OracleBackend

CASES = GC.all_cases()
BACKEND = OracleBackend()


@pytest.mark.parametrize("key,method,kwargs", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_reference_golden(golden, key, method, kwargs):
    out = getattr(BACKEND, method)(**kwargs)
    exp = golden[key]
    assert out.shape == exp.shape
    diff = np.abs(out.astype(np.int16) - exp.astype(np.int16))
    assert diff.max() == 0, f"{key}: {int((diff.max(-1) > 0).sum())} px differ, max channel diff {int(diff.max())}"


def test_golden_coverage(golden):
    """Every golden image the reference's tests hold is exercised by a known-answer case: nothing is left unpinned."""
    covered = {c[0] for c in CASES}
    missing = set(golden.files) - covered
    assert not missing, missing
