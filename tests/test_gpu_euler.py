"""GPU parity (real MI355X, through the C-ABI): Euler residuals — central flux +
scalar JST / matrix JST / Roe-upwind dissipation — and the time step, against
the reference's own Fortran (oracle/_ref)."""
import pytest

import checks
from adflow_amd.params import (FlowParams, dissScalar, dissMatrix, upwind, noLimiter, vanAlbeda, minmod)

pytestmark = pytest.mark.gpu

# BASELINE configs 1-2 parity sizes + ragged / degenerate sizes
SIZES = [(96, 32, 2), (16, 14, 9), (5, 3, 1), (67, 6, 5), (1, 1, 1)]


@pytest.mark.parametrize("dims", SIZES)
def test_euler_scalar_block_res(engine, dims):
    checks.check_block_res(engine, dims, FlowParams(spaceDiscr=dissScalar), seed=sum(dims))


@pytest.mark.parametrize("dims", [(24, 20, 10), (7, 5, 3)])
def test_euler_matrix_block_res(engine, dims):
    checks.check_block_res(engine, dims, FlowParams(spaceDiscr=dissMatrix, vis4=0.1), seed=sum(dims))


@pytest.mark.parametrize("lim", [vanAlbeda, minmod, noLimiter, 1])     # 1 = firstOrder limiter on the fine grid
def test_euler_upwind_block_res(engine, lim):
    checks.check_block_res(engine, (24, 20, 10), FlowParams(spaceDiscr=upwind, limiter=lim), seed=lim)


def test_euler_wall_and_noflux_porosity(engine):
    # boundFlux on the k-min face exercises the wall branch of every scheme
    for sd in (dissScalar, dissMatrix, upwind):
        checks.check_block_res(engine, (12, 10, 8), FlowParams(spaceDiscr=sd), seed=sd, wall_kmin=True)


@pytest.mark.parametrize("sd", [dissScalar, dissMatrix, upwind])
def test_rk_stage_residuals_persistent_fw(engine, sd):
    checks.check_rk_residual_sequence(engine, (14, 12, 6), FlowParams(spaceDiscr=sd))


def test_full_size_block_vs_reference(engine):
    """BASELINE roofline-size block (128^3) against the reference itself."""
    checks.check_block_res(engine, (128, 128, 128), FlowParams(spaceDiscr=dissScalar), seed=5)


@pytest.mark.parametrize("sd", [dissScalar, dissMatrix, upwind])
def test_blanked_cells_and_noflux_faces(engine, sd):
    """iblank = 0 / -1 (overset holes, fringes) and noFlux porosity (conservative
    non-matching boundary): residual_block's max(iblank,0) and porFlux = 0 paths."""
    checks.check_block_res(engine, (33, 12, 9), FlowParams(spaceDiscr=sd), seed=40 + sd, holes=0.08, noflux_jmax=True,
                           wall_kmin=True)


@pytest.mark.parametrize("kch", [4, 5, 64])
def test_euler_march_variants(engine, kch):
    """marching kernel: k-chunk boundaries"""
    engine.set_tuning("march_kch", kch)
    try:
        checks.check_block_res(engine, (70, 9, 11), FlowParams(spaceDiscr=dissScalar), seed=77, wall_kmin=True)
        checks.check_rk_residual_sequence(engine, (20, 6, 7), FlowParams(spaceDiscr=dissScalar), seed=78)
    finally:
        engine.set_tuning("march_kch", 32)


def test_moving_blocks(engine):
    """grid velocities sFaceI/J/K and the rotational source of a steadily rotating block in the central flux, matrix /
    Roe dissipation, spectral radii, SA advection + DDADI coefficients and the D-ADI diagonals
    (fluxes.F90:50,372-397,616,2420; solverUtils.F90:147-181; turbUtils.F90:906; residuals.F90:1192)"""
    from adflow_amd.params import (RANSEquations, DADI, noResAveraging, alternateResAveraging, secondOrder, dissScalar, dissMatrix,
                                   upwind)
    from adflow_amd.topology import BrickTopology
    mv = dict(moving=True)
    for sd in (dissScalar, dissMatrix, upwind):
        checks.check_block_res(engine, (70, 9, 8), FlowParams(spaceDiscr=sd), seed=sd, wall_kmin=True, **mv)
    checks.check_rk_residual_sequence(engine, (24, 10, 8), FlowParams(), **mv)
    checks.check_block_res(engine, (24, 10, 8), FlowParams(equations=RANSEquations, orderTurb=secondOrder), seed=4, stretch_k=2.0, **mv)
    checks.check_rk_smoother(engine, BrickTopology(2, 1, 1, 20, 9, 8), FlowParams(resAveraging=alternateResAveraging), **mv)
    checks.check_dadi_smoother(engine, BrickTopology(1, 2, 1, 16, 8, 8),
                               FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging), stretch_k=2.0, **mv)
    checks.check_sa_solve(engine, BrickTopology(1, 1, 2, 12, 8, 8), FlowParams(equations=RANSEquations, nSubIterTurb=2),
                          stretch_k=2.0, **mv)
    checks.check_mg_cycle(engine, BrickTopology(1, 1, 1, 16, 8, 8), FlowParams(), [0, 1, 0, -1], ncycles=1, **mv)


def test_actuator_regions(engine):
    """a8: actuator-zone source terms in `residual` and after the blocketteRes core"""
    from adflow_amd.params import RANSEquations
    checks.check_actuator_regions(engine, (70, 9, 8), FlowParams())
    checks.check_actuator_regions(engine, (24, 10, 8), FlowParams(equations=RANSEquations, spaceDiscr=upwind), stretch_k=2.0)
    checks.check_actuator_regions(engine, (20, 10, 8), FlowParams(), holes=0.1)


def test_euler_radii_inside_the_march(engine):
    """blocketteRes with its default flags on Euler + scalar JST: the marching kernel forms the spectral radii itself (no k_time_step
    pass).  Against the reference's default path blocketteResCore; partial tiles in i / j, two k chunks, a one-cell-thick block,
    no directional scaling, and the separate-kernel path (updateIntermed = T) on the same inputs."""
    from adflow_amd.params import dissScalar
    for dims in ((63, 9, 35), (124, 6, 5), (16, 8, 1)):
        checks.check_block_res_vs_blockette(engine, dims, FlowParams(spaceDiscr=dissScalar), False, seed=sum(dims),
                                            holes=0.05 if dims[2] > 1 else 0.0)
    # dirScaling = F: blocketteResCore scales unconditionally (blockette.F90:2021-2039), blockResCore honours the switch
    from oracle import ref
    from util import owned, rel_err, TOL
    from adflow_amd.synth import make_block
    prm = FlowParams(spaceDiscr=dissScalar, dirScaling=False)
    engine.release_all()
    blk = make_block(20, 13, 40, prm, seed=73, holes=0.05)
    r = checks.ref_bind(blk, prm)
    ref.block_res_core(False, True, False)
    engine.set_options(prm)
    engine.register(blk)
    engine.blocketteRes(1, False, True, False)
    dw = engine.download_residual()
    assert rel_err(owned(blk, dw), owned(blk, r["dw"])) <= TOL
    # with updateIntermed the radii are an output: the separate k_time_step pass in front of the march
    checks.check_block_res_vs_blockette(engine, (63, 9, 35), FlowParams(spaceDiscr=dissScalar), True, seed=3)
