"""GPU parity (real MI355X, through the C-ABI): the Newton-Krylov residual
function FormFunction_mf = setW + blocketteRes + setRVec (BASELINE config 5;
src/NKSolver/NKSolvers.F90:437-461) on multi-block bricks, and a matrix-free
matvec proxy (finite difference of the residual along a random direction)."""
import numpy as np
import pytest

import checks
from adflow_amd.params import FlowParams, RANSEquations, upwind, dissMatrix
from adflow_amd.topology import BrickTopology

pytestmark = pytest.mark.gpu


def test_nk_residual_euler(engine):
    checks.check_nk_residual(engine, BrickTopology(2, 2, 1, 12, 10, 8), FlowParams())


@pytest.mark.parametrize("sd", [upwind, dissMatrix])
def test_nk_residual_crm_rans_parity_size(engine, sd):
    # BASELINE configs 4/5 parity size: 8 blocks of 24x20x10, RANS-SA, 4a upwind / 4b matrix
    prm = FlowParams(equations=RANSEquations, spaceDiscr=sd, vis4=0.1 if sd == dissMatrix else 0.0156)
    checks.check_nk_residual(engine, BrickTopology(2, 2, 2, 24, 20, 10), prm, stretch_k=3.0)


def test_nk_residual_vector_stores_of_the_marches(engine):
    """setRVec inside the marching kernels: blocks wider than one 60-column tile with a partial last tile, several k chunks,
    RANS (nw = 6: word 5 of a cell comes from the SA march), laminar and Euler Roe (nw = 5)"""
    from adflow_amd.params import NSEquations
    checks.check_nk_residual(engine, BrickTopology(2, 1, 1, 70, 9, 37), FlowParams(equations=RANSEquations, spaceDiscr=upwind), stretch_k=2.0)
    checks.check_nk_residual(engine, BrickTopology(1, 1, 2, 121, 6, 5), FlowParams(equations=NSEquations, spaceDiscr=upwind), stretch_k=2.0)
    checks.check_nk_residual(engine, BrickTopology(1, 2, 1, 63, 5, 4), FlowParams(spaceDiscr=upwind))


@pytest.mark.parametrize("joint", [0, 1])
def test_nk_residual_vector_entries_of_a_cell_by_one_kernel(engine, joint):
    """tuning rvec_joint: the Roe march writes the turbulence entry of the matrix-free vector with its own five, from the dw(itu1) the
    SA march left (1, the default: 48 contiguous bytes per cell), or the SA march writes it itself (0).  A periodic brick of two blocks
    with several tiles and k chunks, and one block with six boundary subfaces (the wall-bounded whole evaluation)"""
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
    try:
        engine.set_tuning("rvec_joint", joint)
        checks.check_nk_residual(engine, BrickTopology(2, 1, 1, 70, 9, 37), rans, stretch_k=2.0)
        checks.check_nk_residual(engine, BrickTopology(1, 1, 1, 13, 9, 6), rans,
                                 bc_spec={1: -6, 2: -6, 3: -1, 4: -6, 5: -3, 6: -6}, stretch_k=2.0)
    finally:
        engine.set_tuning("rvec_joint", 1)


def test_nk_residual_with_floored_pressures(engine):
    """every fifth cell of the vector has less total than kinetic energy: computePressureSimple floors p, whalo2 exchanges the vector's
    energy and recomputes the owned one afterwards (the device pass that does it runs only in this case)"""
    checks.check_nk_residual(engine, BrickTopology(2, 2, 1, 12, 10, 8), FlowParams(), floor_p=True)
    checks.check_nk_residual(engine, BrickTopology(2, 1, 2, 24, 9, 10), FlowParams(equations=RANSEquations, spaceDiscr=dissMatrix, vis4=0.1),
                             floor_p=True, stretch_k=3.0)


def test_matrix_free_matvec_is_linear(engine):
    """(R(w + h v) - R(w)) / h is linear in v to O(h): the property PETSc's MFFD relies on."""
    prm = FlowParams()
    topo = BrickTopology(2, 1, 1, 10, 8, 6)
    blocks, _ = checks.setup_brick(engine, topo, prm, 3)
    w0 = np.concatenate([np.ascontiguousarray(np.transpose(blocks[nn].owned("w"), (2, 1, 0, 3))).reshape(-1)
                         for nn in sorted(blocks)])
    rng = np.random.default_rng(0)
    v1, v2 = rng.standard_normal(w0.size), rng.standard_normal(w0.size)
    v1 /= np.linalg.norm(v1)
    v2 /= np.linalg.norm(v2)
    h = 1e-7
    r0 = engine.FormFunction_mf(w0)
    j1 = (engine.FormFunction_mf(w0 + h * v1) - r0) / h
    j2 = (engine.FormFunction_mf(w0 + h * v2) - r0) / h
    j12 = (engine.FormFunction_mf(w0 + h * (v1 + v2)) - r0) / h
    assert np.linalg.norm(j12 - j1 - j2) <= 1e-5 * np.linalg.norm(j12)
