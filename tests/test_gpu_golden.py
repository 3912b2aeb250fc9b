"""GPU parity against the COMMITTED golden vectors (tests/golden/*.npz, produced
by the reference's own Fortran): independent of oracle/_ref being present."""
import numpy as np
import pytest

from golden_cases import CASES, load_case
from util import TOL, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(CASES))
def test_block_res_vs_golden(engine, name):
    prm, blk, gold, turb = load_case(name)
    engine.release_all()
    engine.set_options(prm)
    engine.register(blk)
    engine.blocketteRes(1, True, True, turb)
    dw = engine.download_residual()
    s = (slice(2, blk.il + 1), slice(2, blk.jl + 1), slice(2, blk.kl + 1))
    for l in range(blk.nw):
        assert rel_err(dw[s][..., l], gold["dw"][..., l]) <= TOL, (name, l)
