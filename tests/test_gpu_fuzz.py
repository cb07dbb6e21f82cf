"""A fixed-seed slice of tools/fuzz_parity.py in the GPU suite: random small planes, contents, modes, qualities, batch
lengths and launch depths, with rollbacks -- the HIP path against the oracle, bit-exact."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_randomised_parity_soak(seed):
    import fuzz_parity
    rng = np.random.default_rng(1000 + seed)
    units = sum(fuzz_parity.one(rng) for _ in range(120))
    assert units > 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_randomised_records_gather_soak(seed):
    """tools/fuzz_parity.py::one_records: random planes in 1..18 bands (even and root-heavy), chunk lengths 1..64, both
    time modes, gray / RGB -- the bands' records expanded by band 0 == the whole-plane stream."""
    import fuzz_parity
    rng = np.random.default_rng(2000 + seed)
    assert sum(fuzz_parity.one_records(rng) for _ in range(60)) > 0
