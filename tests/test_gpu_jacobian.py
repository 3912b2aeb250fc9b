"""GPU parity (real MI355X, through the C-ABI): the coloured finite-difference matrix blocks of
adjointUtils::setupStateResidualMatrix(useAD = F) (adjointUtils.F90:7-715) -- the preconditioner of NK / ANK and the
adjoint's dR/dw -- against the loop nest restated around the reference's own Fortran routines (oracle/_ref, ref_fd_jacobian).
SURVEY.md section 8(f) #4."""
import pytest

import checks
from adflow_amd.params import (FlowParams, dissScalar, dissMatrix, upwind, NSEquations, RANSEquations, secondOrder, vanAlbeda, minmod)

pytestmark = pytest.mark.gpu

WALL = {1: -6, 2: -6, 3: -1, 4: -1, 5: -3, 6: -6}
EULER = {1: -6, 2: -6, 3: -5, 4: -15, 5: -1, 6: -9}
OPEN = {1: -6, 3: -1, 4: -1, 6: -6}          # faces 2 and 5 without subfaces: halos stay as given, like 1-to-1 block interfaces


@pytest.mark.parametrize("sd", [dissScalar, dissMatrix, upwind])
def test_pc_euler(engine, sd):
    checks.check_fd_jacobian(engine, (12, 9, 7), FlowParams(spaceDiscr=sd, limiter=vanAlbeda), EULER)


@pytest.mark.parametrize("sd", [dissScalar, dissMatrix, upwind])
def test_pc_rans(engine, sd):
    """the matrix FormJacobianNK / FormJacobianANK assemble on the north-star discretisations"""
    rans = FlowParams(equations=RANSEquations, spaceDiscr=sd, limiter=vanAlbeda, orderTurb=secondOrder, acousticScaleFactor=0.5,
                      vis4=0.1 if sd == dissMatrix else 0.0156)
    checks.check_fd_jacobian(engine, (16, 10, 7), rans, WALL, stretch_k=2.0)
    checks.check_fd_jacobian(engine, (9, 8, 6), rans, OPEN, stretch_k=2.0)


def test_pc_rans_variants(engine):
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=minmod)
    checks.check_fd_jacobian(engine, (10, 7, 6), rans, WALL, frozenTurb=True, stretch_k=2.0)       # ANK, decoupled
    checks.check_fd_jacobian(engine, (10, 7, 6), rans, WALL, useTurbOnly=True, stretch_k=2.0)      # the turbulence KSP of ANK
    checks.check_fd_jacobian(engine, (8, 7, 6), rans.replace(spaceDiscr=dissScalar), WALL, viscPC=True, stretch_k=2.0)
    checks.check_fd_jacobian(engine, (10, 7, 6), rans, WALL, blockettes=True, stretch_k=2.0)       # blocketteResCore as the evaluator


def test_pc_on_the_kernels_behind_the_marches(engine):
    """the preconditioner matrices again with the marching kernels of the approximate residual switched off one layer at a time
    (tuning pc_fused / roe_march / inviscid_march): k_visc_approx_march + k_roe_march<first order>, then k_inviscid_march<upwind>
    with the first-order limiter, then the gather kernels -- the code paths blocks outside the tile table, moving blocks and coarse
    levels take"""
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=vanAlbeda)
    steps = (("pc_fused", 0), ("roe_march", 0), ("inviscid_march", 0))
    try:
        for key, val in steps:
            engine.set_tuning(key, val)
            checks.check_fd_jacobian(engine, (10, 7, 6), rans, WALL, stretch_k=2.0)
            checks.check_ad_jacobian(engine, (8, 6, 5), rans, WALL, stretch_k=2.0)
        for sd in (dissScalar, dissMatrix):       # their marches are off now: the gather forms of the lumped dissipation
            checks.check_fd_jacobian(engine, (8, 7, 6), rans.replace(spaceDiscr=sd, vis4=0.1 if sd == dissMatrix else 0.0156), WALL, stretch_k=2.0)
    finally:
        engine.set_tuning("pc_fused", 1)
        engine.set_tuning("roe_march", 1)
        engine.set_tuning("inviscid_march", 2)


def test_exact_drdw(engine):
    """usePC = F: 13 colours (Euler) / 35 colours (viscous), 13- and 33-point stencils"""
    checks.check_fd_jacobian(engine, (9, 8, 7), FlowParams(spaceDiscr=dissScalar), EULER, usePC=False)
    checks.check_fd_jacobian(engine, (9, 8, 7), FlowParams(spaceDiscr=upwind, limiter=vanAlbeda), OPEN, usePC=False)
    checks.check_fd_jacobian(engine, (8, 7, 6), FlowParams(equations=NSEquations, spaceDiscr=dissMatrix, vis4=0.1), WALL, usePC=False,
                             stretch_k=2.0)
    checks.check_fd_jacobian(engine, (8, 7, 6), FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=vanAlbeda), WALL,
                             usePC=False, stretch_k=2.0)


# ---- forward-mode assembly (useAD = T, adjointUtils.F90:227-409) against the reference's own Tapenade routines ---------------------
FAR = {1: -6, 2: -6, 3: -5, 4: -6, 5: -1, 6: -6}      # Euler: farfield, an inviscid wall, a symmetry plane


@pytest.mark.parametrize("sd", [dissScalar, dissMatrix, upwind])
def test_ad_pc(engine, sd):
    """the preconditioner matrix by forward mode (ADPC): Euler and RANS-SA, every discretisation.  The Euler case keeps the
    extrapolation / supersonic-outflow faces: applyAllBC_block_d does not linearise them (BCExtra_d.F90) and neither does the
    library (their halos keep value and seed)"""
    checks.check_ad_jacobian(engine, (12, 9, 7), FlowParams(spaceDiscr=sd, limiter=vanAlbeda), EULER)
    rans = FlowParams(equations=RANSEquations, spaceDiscr=sd, limiter=vanAlbeda, orderTurb=secondOrder, acousticScaleFactor=0.5,
                      vis4=0.1 if sd == dissMatrix else 0.0156)
    checks.check_ad_jacobian(engine, (12, 8, 6), rans, WALL, stretch_k=2.0)


@pytest.mark.parametrize("sd", [dissScalar, dissMatrix, upwind])
def test_ad_exact_drdw(engine, sd):
    """the exact dR/dw by forward mode (the adjoint's matrix): 13 / 35 colours, spectral radii, pressure / entropy sensors, limiters,
    nodal gradients and the full viscous flux linearised.  (No extrapolation faces here: a linear extrapolation puts the pressure
    sensor of the boundary cells exactly on the kink of abs(), where the sign of the derivative is the sign of a rounding error)"""
    checks.check_ad_jacobian(engine, (9, 8, 7), FlowParams(spaceDiscr=sd, limiter=vanAlbeda), FAR, usePC=False)
    rans = FlowParams(equations=RANSEquations, spaceDiscr=sd, limiter=vanAlbeda, vis4=0.1 if sd == dissMatrix else 0.0156)
    checks.check_ad_jacobian(engine, (8, 7, 6), rans, WALL, usePC=False, stretch_k=2.0)


def test_ad_variants(engine):
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=minmod)
    checks.check_ad_jacobian(engine, (10, 7, 6), rans, WALL, frozenTurb=True, stretch_k=2.0)
    checks.check_ad_jacobian(engine, (10, 7, 6), rans, WALL, useTurbOnly=True, stretch_k=2.0)
    checks.check_ad_jacobian(engine, (8, 7, 6), rans.replace(spaceDiscr=dissScalar), WALL, viscPC=True, stretch_k=2.0)
    checks.check_ad_jacobian(engine, (8, 7, 6), FlowParams(equations=NSEquations, spaceDiscr=dissMatrix, vis4=0.1), WALL, usePC=False,
                             stretch_k=2.0)
    checks.check_ad_jacobian(engine, (9, 8, 6), rans.replace(limiter=vanAlbeda, useQCR=True), OPEN, usePC=False, stretch_k=2.0)


@pytest.mark.parametrize("sd", [upwind, dissMatrix, dissScalar])
def test_ad_exact_drdw_marches_against_gather_kernels(engine, sd):
    """round 6: the exact linearisation on the MARCHING kernels compiled for dual numbers -- k_visc_gf (gradients + full viscous flux,
    QCR) in front of k_roe_march (second order, van Albada / minmod) or of k_inviscid_march (matrix dissipation, scalar JST) -- against
    the dual gather kernels it replaces (tuning pc_fused = 0; those are checked against the reference's Tapenade routines on small blocks
    in test_ad_exact_drdw): a block of several tiles in i, partial tiles in j and k, several k chunks.  Both are exact derivatives:
    they agree to rounding"""
    import numpy as np
    from adflow_amd.params import minmod
    for lim, qcr, dims in ((vanAlbeda, False, (70, 10, 21)), (minmod, True, (63, 7, 9))):
        rans = FlowParams(equations=RANSEquations, spaceDiscr=sd, limiter=lim, useQCR=qcr, vis4=0.1 if sd == dissMatrix else 0.0156)
        checks.setup_block_with_bc(engine, dims, rans, WALL, 107, stretch_k=2.0)
        try:
            engine.set_tuning("pc_fused", 0)
            engine.setupStateResidualMatrix(1, False, useAD=True)
            Jg = engine.jacobianBlocks(1, 1).copy()
        finally:
            engine.set_tuning("pc_fused", 1)
        engine.setupStateResidualMatrix(1, False, useAD=True)
        Jm = engine.jacobianBlocks(1, 1)
        scale = np.abs(Jg).max()
        assert scale > 0.0
        err = np.abs(Jm - Jg).max() / scale
        assert err <= 2e-10, (lim, qcr, dims, err)
        engine.releaseWorkspace()


def test_ad_agrees_with_finite_differences(engine):
    """the two assemblies of the library against each other on a 24 x 16 x 12 RANS block (several workgroups per direction): the
    forward-mode blocks equal the finite-difference ones to the truncation error of the difference"""
    import numpy as np
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=vanAlbeda)
    checks.setup_block_with_bc(engine, (24, 16, 12), rans, WALL, 105, stretch_k=2.0)
    engine.setupStateResidualMatrix(1, True, useAD=True)
    Ja = engine.jacobianBlocks(1, 1).copy()
    engine.setupStateResidualMatrix(1, True, delta=1e-6)
    Jf = engine.jacobianBlocks(1, 1)
    assert np.abs(Ja[..., :5, :5, :] - Jf[..., :5, :5, :]).max() <= 1e-5 * np.abs(Ja[..., :5, :5, :]).max()
    # round-5 advisor: the slab of dual arrays a forward-mode assembly keeps is handed back on request (640 B per box cell here) and
    # laid out again by the next assembly, whose blocks are the same (to rounding: the finite-difference assembly in between restored
    # the state through the energy, p -> E -> p)
    nbytes = engine.releaseWorkspace()
    assert nbytes >= 600 * (24 + 5) * (16 + 5) * (12 + 5), nbytes
    assert engine.releaseWorkspace() == 0
    engine.setupStateResidualMatrix(1, True, useAD=True)
    assert np.abs(engine.jacobianBlocks(1, 1) - Ja).max() <= 1e-10 * np.abs(Ja).max()


@pytest.mark.parametrize("fused,snap", [(0, 1), (1, 1), (1, 0)])
def test_pc_march_fused_and_pair(engine, fused, snap):
    """tuning pc_fused: the mean-flow residual of the preconditioner matrix on the upwind scheme as ONE march (k_pc_march: first-order
    Roe + thin-layer viscous flux; plain in the finite-difference assembly, on dual numbers in the forward-mode one) or as the kernels
    it replaced (0: k_visc_approx_march + k_roe_march<first order> / the dual gather kernels).  RANS and laminar, a block of several
    tiles in every direction with partial tiles, both assemblies against the reference's.  jac_snap: the marches write the snapshot
    entries of a coloured evaluation themselves (1) or leave dw to k_fd_snap / k_ad_snap (0); frozenTurb / useTurbOnly: only one of the
    two marches runs.  k_pc_march evaluates every j face once (the flux handed to the row above, the cell completed a plane later);
    ny = 9 / 10: tiles whose last rows lie beyond the block, where the wave of the fifth face may be one of them"""
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=vanAlbeda)
    try:
        engine.set_tuning("pc_fused", fused)
        engine.set_tuning("jac_snap", snap)
        checks.check_fd_jacobian(engine, (70, 9, 37), rans, WALL, stretch_k=2.0)
        checks.check_fd_jacobian(engine, (7, 10, 5), rans.replace(equations=NSEquations), OPEN, stretch_k=2.0)
        checks.check_fd_jacobian(engine, (10, 7, 6), rans, WALL, frozenTurb=True, stretch_k=2.0)
        checks.check_fd_jacobian(engine, (10, 7, 6), rans, WALL, useTurbOnly=True, stretch_k=2.0)
        checks.check_ad_jacobian(engine, (10, 7, 6), rans, WALL, useTurbOnly=True, stretch_k=2.0)
        checks.check_fd_jacobian(engine, (11, 9, 6), rans.replace(equations=NSEquations), WALL, stretch_k=2.0)
        checks.check_ad_jacobian(engine, (13, 10, 7), rans, WALL, stretch_k=2.0)
        checks.check_ad_jacobian(engine, (11, 9, 6), rans.replace(equations=NSEquations), OPEN, stretch_k=2.0)
        checks.check_ad_jacobian(engine, (10, 7, 6), rans.replace(limiter=minmod), WALL, frozenTurb=True, stretch_k=2.0)
    finally:
        engine.set_tuning("pc_fused", 1)
        engine.set_tuning("jac_snap", 1)


def test_pc_assemblies_on_a_level_of_several_blocks(engine):
    """three blocks of different sizes in the slots 1, 2, 4: the level-batched marches of the preconditioner assemblies and the
    per-slot snapshot table (checks.check_jacobian_several_blocks)"""
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=vanAlbeda)
    checks.check_jacobian_several_blocks(engine, rans, {
        1: ((24, 8, 6), {1: -6, 2: -6, 3: -1, 4: -4, 5: -3, 6: -6}, {5: -6}),
        2: ((66, 6, 5), {1: -6, 2: -6, 3: -3, 4: -6, 5: -1, 6: -6}, ()),
        4: ((10, 13, 9), {1: -6, 2: -15, 3: -3, 4: -6, 5: -1, 6: -1}, ())}, stretch_k=2.0)


def test_ad_pc_equal_states_across_a_face(engine):
    """extrapolated halos (supersonic outflow, -9) hold the state of the cell behind them: both states of the boundary face are EQUAL,
    eta of the Roe entropy fix is exactly zero and its reciprocal is taken of (z1l + z1r) 1e-290 -- the derivative of the fast dual
    reciprocal must not pass through r^2 (1e580).  Found by tests/fuzz_parity.py --jac (seed 3, case 1): NaN in every block"""
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=minmod, orderTurb=secondOrder, useft2SA=True)
    spec = {1: -9, 2: -3, 3: -1, 4: -3, 5: -1, 6: -9}
    checks.check_ad_jacobian(engine, (5, 6, 4), rans, spec, seed=284201, stretch_k=2.0)
    checks.check_fd_jacobian(engine, (5, 6, 4), rans, spec, seed=284201, stretch_k=2.0)


def test_reference_step(engine):
    """delta = 1e-9 as the reference: rounding differences of two correct residuals are amplified by 1e9, so only ~1e-6 of the
    largest entry is resolvable by ANY implementation (the reference against itself with another compiler flag included)"""
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=vanAlbeda)
    checks.check_fd_jacobian(engine, (12, 8, 6), rans, WALL, delta=1e-9, tol=1e-5, stretch_k=2.0)


def test_pc_rans_larger_block(engine):
    """the preconditioner matrix on a 70 x 24 x 40 block: two i tiles, six j tiles and two k chunks of the marching kernels of the
    approximate residual, the scatter through the LDS column with partial rows"""
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=vanAlbeda)
    checks.check_fd_jacobian(engine, (70, 24, 40), rans, WALL, stretch_k=2.0)
