"""CPU-only: pins the oracles.  (1) the plain-C restatement (oracle/adflow_oracle.c)
against the committed golden vectors, which were produced by the reference's own
Fortran; (2) when oracle/_ref is present, the golden vectors and the restatement
against the reference run live."""
import numpy as np
import pytest

from golden_cases import CASES, load_case
from util import TOL, rel_err

EULER_SCALAR = [n for n in CASES if n.startswith("euler_scalar")]


@pytest.mark.parametrize("name", EULER_SCALAR)
def test_c_restatement_vs_golden(name):
    from oracle import c_oracle
    prm, blk, gold, _ = load_case(name)
    c_oracle.block_res_euler_scalar(blk, prm)
    s = (slice(2, blk.il + 1), slice(2, blk.jl + 1), slice(2, blk.kl + 1))
    for l in range(5):
        assert rel_err(blk["dw"][s][..., l], gold["dw"][..., l]) <= TOL, (name, l)
    for n in ("radI", "radJ", "radK"):
        assert rel_err(blk[n], gold[n]) <= TOL, (name, n)
    assert rel_err(blk["dtl"][1:-1, 1:-1, 1:-1], gold["dtl"]) <= TOL


@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_vs_reference_live(name):
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    prm, blk, gold, turb = load_case(name)
    ref.bind_block(blk, prm)
    ref.block_res_core(True, True, turb)
    s = (slice(2, blk.il + 1), slice(2, blk.jl + 1), slice(2, blk.kl + 1))
    assert np.array_equal(blk["dw"][s], gold["dw"]), name


def test_c_restatement_vs_reference_random_sizes():
    from oracle import c_oracle, ref
    if not ref.available():
        pytest.skip("oracle/_ref not built")
    from adflow_amd.params import FlowParams
    from adflow_amd.synth import make_block
    for dims, seed in (((7, 3, 2), 1), ((1, 1, 1), 2), ((20, 11, 5), 3)):
        prm = FlowParams()
        a = make_block(*dims, prm, seed=seed)
        b = a.copy()
        c_oracle.block_res_euler_scalar(a, prm)
        ref.bind_block(b, prm)
        ref.block_res_core(True, True, False)
        s = (slice(2, a.il + 1), slice(2, a.jl + 1), slice(2, a.kl + 1))
        for l in range(5):
            assert rel_err(a["dw"][s][..., l], b["dw"][s][..., l]) <= TOL


@pytest.mark.parametrize("name", sorted(CASES))
def test_blockette_twin_agrees_with_block_path(name):
    """blockette::blocketteResCore (the reference's default, cache-blocked residual path, blockette.F90:299-753) and the
    blockResCore sequence the golden vectors were generated with must agree to round-off on every golden case (SURVEY.md
    Appendix C: "a free cross-check for the oracle"); the golden vectors therefore pin BOTH twins."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    prm, blk, gold, turb = load_case(name)
    ref.bind_block(blk, prm)
    ref.blockette_res_core(False, True, turb)
    s = (slice(2, blk.il + 1), slice(2, blk.jl + 1), slice(2, blk.kl + 1))
    for l in range(blk.nw):
        assert rel_err(blk["dw"][s][..., l], gold["dw"][..., l]) <= 1e-13, (name, l)
