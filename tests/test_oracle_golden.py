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
    """Every golden of the hot-path categories we claim is exercised (names listed explicitly so a
    missing case is visible)."""
    covered = {c[0] for c in CASES}
    for cat in ("blend", "tools", "scripting", "filters", "adjustments"):
        missing = {k for k in golden.files if k.startswith(cat + "/")} - covered
        # flips are image-crate transforms (out of scope)
        missing -= {"scripting/flip_horizontal", "scripting/flip_vertical"}
        assert not missing, missing
