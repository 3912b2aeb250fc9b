"""CPU-only CI: the SAME kernel sources as the GPU library, compiled with g++
against tests/hostsim (an emulation of the HIP launch/barrier/shuffle surface),
checked against the reference's own Fortran.  This validates kernel arithmetic
and indexing without a GPU; the -m gpu suite repeats the checks on the MI355X."""
import pytest

import checks
from adflow_amd.params import (FlowParams, dissScalar, dissMatrix, upwind, noLimiter, vanAlbeda, minmod, NSEquations,
                               RANSEquations, secondOrder, vorticity, DADI)
from oracle import ref

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")


@pytest.mark.parametrize("dims", [(16, 14, 9), (5, 3, 1), (1, 1, 1)])
def test_euler_scalar(hostsim_engine, dims):
    checks.check_block_res(hostsim_engine, dims, FlowParams(spaceDiscr=dissScalar), seed=sum(dims))


def test_euler_matrix(hostsim_engine):
    checks.check_block_res(hostsim_engine, (12, 10, 6), FlowParams(spaceDiscr=dissMatrix, vis4=0.1), seed=2)


@pytest.mark.parametrize("lim", [vanAlbeda, minmod, noLimiter, 1])     # 1 = firstOrder limiter on the fine grid
def test_euler_upwind(hostsim_engine, lim):
    checks.check_block_res(hostsim_engine, (12, 10, 6), FlowParams(spaceDiscr=upwind, limiter=lim), seed=lim)


def test_wall_porosity(hostsim_engine):
    for sd in (dissScalar, dissMatrix, upwind):
        checks.check_block_res(hostsim_engine, (8, 6, 5), FlowParams(spaceDiscr=sd), seed=sd, wall_kmin=True)


@pytest.mark.parametrize("sd", [dissScalar, dissMatrix, upwind])
def test_rk_stage_residuals(hostsim_engine, sd):
    checks.check_rk_residual_sequence(hostsim_engine, (10, 8, 6), FlowParams(spaceDiscr=sd))


def test_laminar_ns(hostsim_engine):
    checks.check_block_res(hostsim_engine, (8, 6, 5), FlowParams(equations=NSEquations), seed=4, stretch_k=2.0)


@pytest.mark.parametrize("sd", [dissScalar, dissMatrix, upwind])
def test_rans_sa(hostsim_engine, sd):
    prm = FlowParams(equations=RANSEquations, spaceDiscr=sd, vis4=0.1 if sd == dissMatrix else 0.0156)
    checks.check_block_res(hostsim_engine, (10, 8, 6), prm, seed=sd, stretch_k=2.5)


def test_rans_sa_options(hostsim_engine):
    prm = FlowParams(equations=RANSEquations, orderTurb=secondOrder, turbProd=vorticity, useQCR=True, useft2SA=False)
    checks.check_block_res(hostsim_engine, (7, 9, 5), prm, seed=8, stretch_k=2.0)


def test_rans_rk_stage_residuals(hostsim_engine):
    checks.check_rk_residual_sequence(hostsim_engine, (8, 6, 5), FlowParams(equations=NSEquations), stretch_k=2.0)


# ---- shell routines: halo exchange and smoothers on periodic bricks -------------
from adflow_amd.params import noResAveraging, alternateResAveraging  # noqa: E402
from adflow_amd.topology import BrickTopology  # noqa: E402


@pytest.mark.parametrize("nLayers", [1, 2])
def test_halo_exchange(hostsim_engine, nLayers):
    checks.check_halo_exchange(hostsim_engine, BrickTopology(2, 2, 1, 5, 4, 3), FlowParams(equations=RANSEquations), nLayers)


@pytest.mark.parametrize("resavg", [noResAveraging, alternateResAveraging])
def test_rk_smoother(hostsim_engine, resavg):
    checks.check_rk_smoother(hostsim_engine, BrickTopology(2, 1, 1, 6, 5, 4), FlowParams(resAveraging=resavg))


def test_res_averaging_line_bundles(hostsim_engine):
    """k_ra_line: more lines than one bundle in every direction (ragged last bundle), blocks of different sizes on one level, a line too
    long for the LDS buffer (two-pass kernels), one-cell directions"""
    from adflow_amd.params import alwaysResAveraging
    prm = FlowParams(resAveraging=alwaysResAveraging)
    checks.check_rk_smoother(hostsim_engine, BrickTopology(1, 1, 1, 19, 18, 3), prm, holes=0.05)
    checks.check_rk_smoother(hostsim_engine, BrickTopology(1, 1, 1, 3, 2, 19), prm)
    checks.check_rk_smoother(hostsim_engine, BrickTopology(1, 1, 1, 1340, 2, 1), prm)
    checks.check_rk_smoother(hostsim_engine, BrickTopology(1, 1, 1, 2, 1040, 1), prm)


def test_rk_smoother_rans(hostsim_engine):
    checks.check_rk_smoother(hostsim_engine, BrickTopology(1, 2, 1, 6, 5, 4),
                             FlowParams(equations=RANSEquations, resAveraging=noResAveraging), stretch_k=2.0)


def test_dadi_smoother(hostsim_engine):
    checks.check_dadi_smoother(hostsim_engine, BrickTopology(2, 1, 1, 6, 5, 4), FlowParams(resAveraging=noResAveraging, cfl=1.5))
    prm = FlowParams(equations=RANSEquations, resAveraging=noResAveraging, cfl=1.5, nSubiterations=3)
    checks.check_dadi_smoother(hostsim_engine, BrickTopology(1, 1, 2, 5, 1, 4), prm, stretch_k=2.0)
    # i lines longer than one 62-cell tile of k_dadi_rows_i (the neighbours' coefficients by lane shifts), a partial row group
    checks.check_dadi_smoother(hostsim_engine, BrickTopology(1, 1, 1, 65, 5, 3), FlowParams(resAveraging=noResAveraging, cfl=1.5), holes=0.05)
    checks.check_dadi_smoother(hostsim_engine, BrickTopology(1, 1, 1, 63, 2, 2), prm, stretch_k=2.0)


from golden_cases import CASES as _GOLD, load_case as _load_case  # noqa: E402
from util import TOL as _TOL, rel_err as _rel_err  # noqa: E402


@pytest.mark.parametrize("name", sorted(_GOLD))
def test_kernels_vs_golden(hostsim_engine, name):
    prm, blk, gold, turb = _load_case(name)
    hostsim_engine.release_all()
    hostsim_engine.set_options(prm)
    hostsim_engine.register(blk)
    hostsim_engine.blocketteRes(1, True, True, turb)
    dw = hostsim_engine.download_residual()
    s = (slice(2, blk.il + 1), slice(2, blk.jl + 1), slice(2, blk.kl + 1))
    for l in range(blk.nw):
        assert _rel_err(dw[s][..., l], gold["dw"][..., l]) <= _TOL, (name, l)


def test_mg_transfer(hostsim_engine):
    checks.check_mg_transfer(hostsim_engine, BrickTopology(2, 1, 1, 8, 6, 4), FlowParams(resAveraging=noResAveraging))


@pytest.mark.parametrize("cycling", [[0, 1, 0, -1], [0, 1, 0, 0, -1, 0]])
def test_mg_cycle(hostsim_engine, cycling):
    checks.check_mg_cycle(hostsim_engine, BrickTopology(2, 1, 1, 8, 6, 4), FlowParams(), cycling)


def test_mg_irregular_coarsening(hostsim_engine):
    """odd cell counts and subface boundaries on even nodes: coarse cells of ONE fine cell with restriction weight 1/2
    (createCoarseBlocks, coarseUtils.F90:117-153, 281-343); twins of tests/test_gpu_multigrid.py::test_mg_*_irregular*"""
    e = hostsim_engine
    checks.check_mg_transfer(e, BrickTopology(2, 1, 1, 9, 7, 5), FlowParams(resAveraging=noResAveraging), irregular=(3, 0))
    checks.check_mg_cycle(e, BrickTopology(1, 1, 1, 11, 9, 7), FlowParams(), [0, 1, 0, 1, 0, -1, 0, 1, 0, -1, 0, -1, 0], ncycles=1,
                          nlevels=3, irregular=(4, 0))
    spec = {1: -6, 2: -6, 3: -5, 4: -6, 5: -1, 6: -1}
    checks.check_mg_cycle(e, BrickTopology(1, 1, 1, 10, 8, 4), FlowParams(), [0, 1, 0, -1], ncycles=1, bc_spec=spec,
                          bc_split={3: -6, 5: -5}, irregular=(1, 1))
    rans = FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging, cfl=1.5, nSubiterations=2, nSubIterTurb=2)
    checks.check_mg_cycle(e, BrickTopology(1, 1, 1, 9, 5, 7), rans, [0, 1, 0, 1, 0, -1, 0, -1], ncycles=1, nlevels=3,
                          bc_spec={1: -6, 2: -6, 3: -1, 4: -1, 5: -3, 6: -6}, bc_split={5: -6}, irregular=(4, 1), stretch_k=2.0)
    checks.check_coarse_level_geometry(e, BrickTopology(2, 1, 1, 9, 7, 5), FlowParams())


def test_foreign_normals_then_own_nodes(hostsim_engine):
    checks.check_foreign_normals_then_own_nodes(hostsim_engine, (9, 7, 6), FlowParams(equations=RANSEquations, spaceDiscr=upwind), stretch_k=2.0)


def test_bc_faces_wider_than_the_edge_rings(hostsim_engine):
    """faces with cells off the block edges (more than 6 cells wide): every kind, split faces, smoother and blocketteRes with subfaces"""
    e = hostsim_engine
    checks.check_apply_bc(e, (12, 10, 9), FlowParams(), {1: -1, 2: -6, 3: -5, 4: -15, 5: -1, 6: -9})
    checks.check_apply_bc(e, (12, 10, 9), FlowParams(equations=RANSEquations), {1: -6, 2: -6, 3: -1, 4: -4, 5: -3, 6: -6}, stretch_k=2.0)
    checks.check_apply_bc(e, (14, 10, 9), FlowParams(), {1: -1, 2: -1, 3: -1, 4: -6, 5: -6, 6: -1}, split={3: -6, 6: -5}, secondHalo=False)
    rans = FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging, cfl=1.5, nSubiterations=2, nSubIterTurb=2)
    checks.check_smoother_with_bc(e, (10, 9, 8), rans, {1: -6, 2: -6, 3: -1, 4: -1, 5: -3, 6: -6}, stretch_k=2.0)
    checks.check_blockette_res_with_bc(e, BrickTopology(2, 1, 1, 10, 9, 8, periodic=(False, False, False)),
                                       FlowParams(equations=RANSEquations), {1: -6, 2: -6, 3: -1, 4: -6, 5: -3, 6: -6}, stretch_k=2.0)


def test_blockette_res_with_bc_on_thin_blocks(hostsim_engine):
    """the whole blocketteRes with boundary subfaces on blocks of one to a few cells in one or several directions; floored pressures"""
    e = hostsim_engine
    spec = {1: -6, 2: -6, 3: -1, 4: -6, 5: -3, 6: -6}
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
    for dims in ((9, 7, 6), (5, 4, 3), (3, 2, 1), (70, 5, 2), (4, 9, 8)):
        checks.check_blockette_res_with_bc(e, BrickTopology(2, 1, 1, *dims, periodic=(False, False, False)), rans, spec, stretch_k=2.0)
    checks.check_blockette_res_with_bc(e, BrickTopology(1, 2, 1, 8, 6, 5, periodic=(False, False, False)), rans, spec, floor_p=True, stretch_k=2.0)


def test_rotated_interfaces(hostsim_engine):
    """1-to-1 interfaces with a transformation between blocks of different sizes (modules/block.F90:271-309): the emulator twin
    of tests/test_gpu_topology.py"""
    from adflow_amd.topology import ell_topology
    e = hostsim_engine
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
    for nLayers in (1, 2):
        checks.check_halo_exchange(e, ell_topology(), rans, nLayers)
        checks.check_halo_loopback(e, ell_topology(), 3, rans, nLayers)
    checks.check_coordinate_halos_brick(e, ell_topology(), FlowParams())
    walls = {1: -6, 2: -6, 3: -1, 4: -6, 5: -3, 6: -6}
    for se in (None, 2):
        assert checks.check_blockette_res_with_bc(e, ell_topology(stretch_z=2.0), rans, walls, split_eval=se) == 2
    spec = {1: -6, 2: -6, 3: -6, 4: -6, 5: -5, 6: -6}
    assert checks.check_pressure_early_exchange(e, None, FlowParams(), topo=ell_topology(), lattice_spec=spec) > 1e-6
    checks.check_dadi_smoother(e, ell_topology(), FlowParams(resAveraging=noResAveraging, cfl=1.5))
    checks.check_mg_cycle(e, ell_topology(), FlowParams(), [0, 1, 0, -1], ncycles=1, brick_spec={1: -6, 2: -6, 3: -1, 4: -6, 5: -5, 6: -6})
    checks.check_nk_residual(e, ell_topology(), rans)


def test_nk_residual(hostsim_engine):
    checks.check_nk_residual(hostsim_engine, BrickTopology(2, 1, 1, 6, 5, 4), FlowParams())
    checks.check_nk_residual(hostsim_engine, BrickTopology(1, 2, 1, 6, 5, 4),
                             FlowParams(equations=RANSEquations, spaceDiscr=upwind), stretch_k=2.0)
    # two column tiles, the second partial: the word-ordered setRVec stores of the Roe march (nw = 6 and nw = 5)
    checks.check_nk_residual(hostsim_engine, BrickTopology(1, 1, 1, 63, 5, 4),
                             FlowParams(equations=RANSEquations, spaceDiscr=upwind), stretch_k=2.0)
    checks.check_nk_residual(hostsim_engine, BrickTopology(1, 1, 1, 61, 3, 3), FlowParams(spaceDiscr=upwind))
    # pressures at the floor: the halos take the energy of the vector, the owned cells the recomputed one
    checks.check_nk_residual(hostsim_engine, BrickTopology(2, 2, 1, 6, 5, 4), FlowParams(), floor_p=True)
    # tuning rvec_joint = 0: the SA march writes the turbulence entry of the vector itself (1, the default, ran above)
    try:
        hostsim_engine.set_tuning("rvec_joint", 0)
        checks.check_nk_residual(hostsim_engine, BrickTopology(1, 2, 1, 6, 5, 4),
                                 FlowParams(equations=RANSEquations, spaceDiscr=upwind), stretch_k=2.0)
    finally:
        hostsim_engine.set_tuning("rvec_joint", 1)


def test_sa_ddadi_solve(hostsim_engine):
    prm = FlowParams(equations=RANSEquations, nSubIterTurb=2, orderTurb=secondOrder)
    checks.check_sa_solve(hostsim_engine, BrickTopology(2, 1, 1, 6, 5, 4), prm, stretch_k=2.0)


def test_mg_cycle_rans_single_grid(hostsim_engine):
    from adflow_amd.params import DADI
    prm = FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging, cfl=1.5, nSubiterations=2,
                     nSubIterTurb=2)
    checks.check_mg_cycle(hostsim_engine, BrickTopology(2, 1, 1, 8, 6, 4), prm, [0], ncycles=1, stretch_k=2.0)


# ---- edge cases the reference handles: blanked (overset) cells, noFlux faces --------
@pytest.mark.parametrize("sd", [dissScalar, dissMatrix, upwind])
def test_blanked_cells_and_noflux_faces(hostsim_engine, sd):
    prm = FlowParams(spaceDiscr=sd)
    checks.check_block_res(hostsim_engine, (9, 7, 6), prm, seed=40 + sd, holes=0.08, noflux_jmax=True, wall_kmin=True)


def test_blanked_cells_rans_and_smoothers(hostsim_engine):
    prm = FlowParams(equations=RANSEquations)
    checks.check_block_res(hostsim_engine, (9, 7, 6), prm, seed=50, holes=0.08, noflux_jmax=True, stretch_k=2.0)
    topo = BrickTopology(2, 1, 1, 6, 5, 4)
    checks.check_rk_smoother(hostsim_engine, topo, FlowParams(), holes=0.1)
    checks.check_dadi_smoother(hostsim_engine, topo, FlowParams(resAveraging=noResAveraging, cfl=1.5), holes=0.1)
    checks.check_sa_solve(hostsim_engine, topo, FlowParams(equations=RANSEquations, nSubIterTurb=2), holes=0.1, stretch_k=2.0)


def test_mg_cycle_three_levels(hostsim_engine):
    checks.check_mg_cycle(hostsim_engine, BrickTopology(1, 1, 1, 8, 8, 8), FlowParams(),
                          [0, 1, 0, 1, 0, -1, 0, 1, 0, -1, 0, -1, 0], ncycles=1, nlevels=3)


@pytest.mark.parametrize("kch", [4, 5, 32])
def test_euler_march_variants(hostsim_engine, kch):
    """marching kernel: chunk boundaries (k-chunks of 4/5 planes on an 11-plane block)"""
    hostsim_engine.set_tuning("march_kch", kch)
    try:
        checks.check_block_res(hostsim_engine, (13, 6, 11), FlowParams(spaceDiscr=dissScalar), seed=77, wall_kmin=True)
        checks.check_rk_residual_sequence(hostsim_engine, (9, 5, 7), FlowParams(spaceDiscr=dissScalar), seed=78)
    finally:
        hostsim_engine.set_tuning("march_kch", 32)


# ---- boundary conditions on the device ("next" row 1): same checks as tests/test_gpu_bc.py, small sizes ----
@pytest.mark.parametrize("spec", [{1: -1, 2: -6, 3: -5, 4: -15, 5: -1, 6: -9}, {1: -7, 2: -6, 3: -5, 4: -5, 5: -1, 6: -1}])
@pytest.mark.parametrize("second", [True, False])
def test_apply_all_bc_euler(hostsim_engine, spec, second):
    checks.check_apply_bc(hostsim_engine, (6, 5, 4), FlowParams(), spec, secondHalo=second)
    checks.check_apply_bc(hostsim_engine, (6, 5, 1), FlowParams(eulerWallBCTreatment=1, outflowTreatment=2), spec, secondHalo=second)


def test_apply_all_bc_viscous(hostsim_engine):
    checks.check_apply_bc(hostsim_engine, (5, 4, 3), FlowParams(equations=RANSEquations), {1: -6, 2: -6, 3: -1, 4: -4, 5: -3, 6: -6},
                          stretch_k=2.0)
    checks.check_apply_bc(hostsim_engine, (5, 4, 3), FlowParams(equations=NSEquations, viscWallBCTreatment=2),
                          {1: -9, 2: -7, 3: -3, 4: -1, 5: -4, 6: -15}, stretch_k=2.0)
    checks.check_apply_bc(hostsim_engine, (6, 4, 4), FlowParams(equations=NSEquations, viscWallBCTreatment=2),
                          {1: -6, 2: -6, 3: -3, 4: -1, 5: -5, 6: -6}, secondHalo=False, level=2, stretch_k=2.0)


def test_smoothers_sa_solve_mg_nk_with_bc(hostsim_engine):
    from adflow_amd.params import DADI, RungeKutta, alternateResAveraging
    e = hostsim_engine
    checks.check_smoother_with_bc(e, (8, 6, 4), FlowParams(smoother=RungeKutta, resAveraging=alternateResAveraging),
                                  {1: -6, 2: -6, 3: -5, 4: -6, 5: -1, 6: -1})
    rans = FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging, cfl=1.5, nSubiterations=2, nSubIterTurb=2)
    wall = {1: -6, 2: -6, 3: -1, 4: -1, 5: -3, 6: -6}
    checks.check_smoother_with_bc(e, (6, 5, 4), rans, wall, stretch_k=2.0)
    checks.check_sa_solve_with_bc(e, (6, 5, 4), rans.replace(orderTurb=secondOrder), {1: -6, 2: -15, 3: -1, 4: -4, 5: -3, 6: -9},
                                  stretch_k=2.0)
    checks.check_mg_cycle(e, BrickTopology(1, 1, 1, 8, 8, 4), FlowParams(), [0, 1, 0, -1],
                          bc_spec={1: -6, 2: -6, 3: -5, 4: -6, 5: -1, 6: -1})
    checks.check_mg_cycle(e, BrickTopology(1, 1, 1, 8, 4, 4), rans, [0, 1, 0, -1], ncycles=1, bc_spec=wall, stretch_k=2.0)
    checks.check_nk_residual(e, BrickTopology(1, 1, 1, 6, 5, 4), FlowParams(equations=RANSEquations), bc_spec=wall, stretch_k=2.0)


def test_update_geometry_after_mesh_warp(hostsim_engine):
    checks.check_update_geometry(hostsim_engine, (7, 5, 4), FlowParams(), {1: -1, 2: -6, 3: -5, 4: -5, 5: -1, 6: -6})
    checks.check_update_geometry(hostsim_engine, (5, 4, 1), FlowParams(equations=NSEquations), {1: -6, 2: -6, 3: -3, 4: -6, 5: -1, 6: -1},
                                 stretch_k=2.0)


def test_apply_all_bc_split_faces(hostsim_engine):
    """block faces cut into two subfaces of different kinds (wall + farfield, symmetry + Euler wall)"""
    checks.check_apply_bc(hostsim_engine, (8, 6, 4), FlowParams(), {1: -1, 2: -6, 3: -5, 4: -15, 5: -1, 6: -9},
                          split={3: -6, 6: -5, 1: -6})
    checks.check_apply_bc(hostsim_engine, (8, 6, 4), FlowParams(equations=RANSEquations), {1: -6, 2: -6, 3: -1, 4: -4, 5: -3, 6: -6},
                          split={5: -6, 4: -3}, stretch_k=2.0)


def test_low_speed_preconditioner(hostsim_engine):
    """residual_block's 5x5 low-Mach transform (residuals.F90:172-331) and the 0.8 RK step factor (smoothers.F90:202);
    blocketteRes does not apply it (blockette.F90:755-852)"""
    lo = dict(lowSpeedPreconditioner=True, Mach=0.15)
    checks.check_rk_residual_sequence(hostsim_engine, (9, 7, 5), FlowParams(**lo))
    checks.check_rk_residual_sequence(hostsim_engine, (8, 6, 5), FlowParams(equations=RANSEquations, spaceDiscr=upwind, **lo), stretch_k=2.0)
    checks.check_rk_smoother(hostsim_engine, BrickTopology(2, 1, 1, 6, 5, 4), FlowParams(resAveraging=alternateResAveraging, **lo))
    checks.check_dadi_smoother(hostsim_engine, BrickTopology(1, 2, 1, 6, 5, 4),
                               FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging, **lo), stretch_k=2.0)
    checks.check_block_res(hostsim_engine, (8, 6, 5), FlowParams(**lo), seed=5)


def test_wall_stress_storage(hostsim_engine):
    """a7 / a17 useStoreWall: viscSubface%tau, %q on all six block faces, split subfaces, QCR"""
    checks.check_wall_stress(hostsim_engine, (6, 5, 4), FlowParams(equations=NSEquations), {1: -3, 2: -4, 3: -3, 4: -6, 5: -4, 6: -3},
                             stretch_k=2.0)
    checks.check_wall_stress(hostsim_engine, (7, 5, 4), FlowParams(equations=RANSEquations, useQCR=True),
                             {1: -6, 2: -6, 3: -1, 4: -4, 5: -3, 6: -6}, split={5: -6, 4: -3}, stretch_k=2.0)


def test_apply_all_bc_subsonic_and_polar(hostsim_engine):
    """symmPolar, subsonic inflow (total conditions on min faces, mass flow on max faces, hScalingInlet), subsonic
    outflow / outflow bleed, with the turbulence inflow / outflow treatment for RANS"""
    lo = dict(Mach=0.3)
    spec = {1: -8, 2: -10, 3: -2, 4: -8, 5: -12, 6: -6}
    for second in (True, False):
        checks.check_apply_bc(hostsim_engine, (6, 5, 4), FlowParams(**lo), spec, secondHalo=second)
    checks.check_apply_bc(hostsim_engine, (6, 5, 4), FlowParams(hScalingInlet=True, **lo), {1: -10, 2: -8, 3: -8, 4: -2, 5: -1, 6: -5})
    rans = FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging, **lo)
    checks.check_multiblock_bc(hostsim_engine, rans, {
        1: ((6, 5, 4), {1: -8, 2: -10, 3: -3, 4: -6, 5: -2, 6: -7}, ()),
        2: ((5, 4, 6), {1: -12, 2: -8, 3: -3, 4: -6}, {3: -6})}, stretch_k=2.0)


def test_moving_blocks(hostsim_engine):
    """grid velocities sFaceI/J/K and the rotational source of a steadily rotating block in the central flux, matrix /
    Roe dissipation, spectral radii, SA advection + DDADI coefficients and the D-ADI diagonals
    (fluxes.F90:50,372-397,616,2420; solverUtils.F90:147-181; turbUtils.F90:906; residuals.F90:1192)"""
    mv = dict(moving=True)
    for sd in (dissScalar, dissMatrix, upwind):
        checks.check_block_res(hostsim_engine, (8, 6, 5), FlowParams(spaceDiscr=sd), seed=sd, wall_kmin=True, **mv)
    checks.check_rk_residual_sequence(hostsim_engine, (9, 7, 5), FlowParams(), **mv)
    checks.check_block_res(hostsim_engine, (8, 6, 5), FlowParams(equations=RANSEquations, orderTurb=secondOrder), seed=4, stretch_k=2.0, **mv)
    checks.check_rk_smoother(hostsim_engine, BrickTopology(2, 1, 1, 6, 5, 4), FlowParams(resAveraging=alternateResAveraging), **mv)
    checks.check_dadi_smoother(hostsim_engine, BrickTopology(1, 2, 1, 6, 5, 4),
                               FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging), stretch_k=2.0, **mv)
    checks.check_sa_solve(hostsim_engine, BrickTopology(1, 1, 2, 5, 4, 4), FlowParams(equations=RANSEquations, nSubIterTurb=2),
                          stretch_k=2.0, **mv)
    checks.check_mg_cycle(hostsim_engine, BrickTopology(1, 1, 1, 8, 8, 4), FlowParams(), [0, 1, 0, -1], ncycles=1, **mv)


def test_coordinate_halos(hostsim_engine):
    """"next" row 3: xhalo_block + exchangeCoor (node pattern) + metrics after a mesh warp"""
    checks.check_coordinate_halos_brick(hostsim_engine, BrickTopology(2, 2, 1, 5, 4, 3), FlowParams())
    checks.check_coordinate_halos_brick(hostsim_engine, BrickTopology(1, 1, 2, 4, 4, 2), FlowParams(equations=NSEquations), stretch_k=2.0)
    checks.check_xhalo_symmetry(hostsim_engine, (6, 5, 4), FlowParams(), {1: -1, 2: -6, 3: -5, 4: -1, 5: -1, 6: -6})
    checks.check_xhalo_symmetry(hostsim_engine, (7, 5, 4), FlowParams(), {1: -1, 2: -1, 3: -1, 4: -6, 5: -6, 6: -1}, split={3: -6, 6: -5})
    checks.check_coarse_level_geometry(hostsim_engine, BrickTopology(2, 1, 1, 8, 6, 4), FlowParams())


def test_periodic_halos(hostsim_engine):
    """a18: periodic transformations of the halo exchange (velocities) and of exchangeCoor (coordinates)"""
    checks.check_periodic_halos(hostsim_engine, BrickTopology(2, 1, 1, 5, 4, 3), FlowParams())
    checks.check_periodic_halos(hostsim_engine, BrickTopology(1, 2, 1, 4, 4, 2), FlowParams(equations=RANSEquations), stretch_k=2.0)


def test_actuator_regions(hostsim_engine):
    """a8: actuator-zone source terms in `residual` and after the blocketteRes core"""
    checks.check_actuator_regions(hostsim_engine, (8, 6, 5), FlowParams())
    checks.check_actuator_regions(hostsim_engine, (7, 6, 5), FlowParams(equations=RANSEquations, spaceDiscr=upwind), stretch_k=2.0)
    checks.check_actuator_regions(hostsim_engine, (7, 6, 5), FlowParams(), holes=0.1)


def test_inviscid_march_variants(hostsim_engine):
    """tuning inviscid_march: 0 = cell-gather kernel for matrix / upwind, 1 = gather kernel for NS / RANS scalar JST (2, the default: marching form there too)"""
    try:
        hostsim_engine.set_tuning("inviscid_march", 0)
        for sd in (dissMatrix, upwind):
            checks.check_block_res(hostsim_engine, (9, 6, 5), FlowParams(equations=RANSEquations, spaceDiscr=sd), seed=sd, stretch_k=2.0)
        hostsim_engine.set_tuning("inviscid_march", 1)
        hostsim_engine.set_tuning("march_kch", 4)
        checks.check_block_res(hostsim_engine, (13, 6, 7), FlowParams(equations=RANSEquations), seed=6, stretch_k=2.0)
        checks.check_rk_residual_sequence(hostsim_engine, (9, 5, 7), FlowParams(equations=NSEquations), stretch_k=2.0)
        hostsim_engine.set_tuning("inviscid_march", 2)          # the default form again, with partial k chunks
        checks.check_block_res(hostsim_engine, (13, 6, 7), FlowParams(equations=RANSEquations), seed=7, stretch_k=2.0)
        checks.check_rk_residual_sequence(hostsim_engine, (9, 5, 7), FlowParams(equations=NSEquations), stretch_k=2.0)
    finally:
        hostsim_engine.set_tuning("inviscid_march", 2)
        hostsim_engine.set_tuning("march_kch", 32)


def test_multiblock_bc(hostsim_engine):
    """several blocks with different subface lists: the level-batched BC launches against the reference's block loop"""
    checks.check_multiblock_bc(hostsim_engine, FlowParams(), {
        1: ((8, 6, 4), {1: -1, 2: -6, 3: -5, 4: -15, 5: -1, 6: -9}, {3: -6, 6: -5}),
        2: ((5, 7, 3), None, ()),
        3: ((6, 4, 5), {2: -6, 3: -5, 6: -7}, ()),
        4: ((4, 4, 4), {1: -5, 2: -5, 3: -5, 4: -5, 5: -5, 6: -5}, {1: -6})})
    checks.check_multiblock_bc(hostsim_engine, FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging), {
        1: ((6, 5, 4), {1: -6, 2: -6, 3: -1, 4: -4, 5: -3, 6: -6}, {5: -6, 4: -3}),
        2: ((5, 4, 6), {3: -3, 4: -6}, ()),
        3: ((4, 6, 4), {1: -6, 2: -9, 3: -3, 4: -6, 5: -1, 6: -1}, ())}, stretch_k=2.0)


# ---- approximate residual of the preconditioner assembly (blocketteRes useDissApprox / useViscApprox) ----
@pytest.mark.parametrize("sd", [dissScalar, dissMatrix, upwind])
def test_block_res_approx(hostsim_engine, sd):
    checks.check_block_res_approx(hostsim_engine, (9, 7, 5), FlowParams(spaceDiscr=sd, sigma=0.2, vis4=0.05), visc_approx=False)
    checks.check_block_res_approx(hostsim_engine, (8, 6, 5), FlowParams(equations=RANSEquations, spaceDiscr=sd, sigma=0.3, vis4=0.05),
                                  stretch_k=2.0)


def test_block_res_approx_blockette_core(hostsim_engine):
    """the default core (blocketteResCore) evaluates the approximate Roe flux first order (blockette.F90:643)"""
    checks.check_block_res_approx(hostsim_engine, (9, 7, 5), FlowParams(spaceDiscr=upwind, limiter=vanAlbeda), visc_approx=False,
                                  blockettes=True)
    checks.check_block_res_approx(hostsim_engine, (8, 6, 5), FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=minmod),
                                  blockettes=True, stretch_k=2.0)
    checks.check_block_res_approx(hostsim_engine, (8, 6, 5), FlowParams(equations=RANSEquations, spaceDiscr=dissScalar, sigma=0.3),
                                  blockettes=True, stretch_k=2.0)


def test_block_res_visc_approx_only(hostsim_engine):
    checks.check_block_res_approx(hostsim_engine, (8, 6, 5), FlowParams(equations=NSEquations, sigma=0.3), diss_approx=False,
                                  visc_approx=True, stretch_k=2.0)


# ---- round 2 kernels on the emulator: viscous-dominated states, kernel variants, default flags (GPU twins in test_gpu_rans.py)
@pytest.mark.parametrize("eq,qcr", [(NSEquations, False), (RANSEquations, True)])
def test_viscous_dominated(hostsim_engine, eq, qcr):
    prm = FlowParams(equations=eq, spaceDiscr=upwind, useQCR=qcr, muSuthDim=1.0)
    checks.check_block_res(hostsim_engine, (63, 6, 9), prm, seed=21, stretch_k=2.0)
    checks.check_rk_residual_sequence(hostsim_engine, (12, 6, 5), FlowParams(equations=NSEquations, muSuthDim=1.0), stretch_k=2.0)


def test_viscous_kernel_variants(hostsim_engine):
    import test_gpu_rans
    test_gpu_rans.test_viscous_kernel_variants.__wrapped__(hostsim_engine) if hasattr(test_gpu_rans.test_viscous_kernel_variants, "__wrapped__") \
        else test_gpu_rans.test_viscous_kernel_variants(hostsim_engine)


def test_visc_gradient_fused(hostsim_engine):
    import test_gpu_rans
    test_gpu_rans.test_visc_gradient_fused(hostsim_engine)


def test_multiblock_brick_block_res(hostsim_engine):
    from adflow_amd.topology import BrickTopology
    prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
    checks.check_brick_block_res(hostsim_engine, BrickTopology(2, 1, 2, 10, 7, 6), prm, seed=3, stretch_k=2.0)


def test_dadi_i_direction_by_cyclic_reduction(hostsim_engine):
    import test_gpu_smoothers
    test_gpu_smoothers.test_dadi_i_direction_by_cyclic_reduction(hostsim_engine)


def test_blockette_res_wall_bounded_brick(hostsim_engine):
    import test_gpu_rans
    test_gpu_rans.test_blockette_res_wall_bounded_brick(hostsim_engine)


def test_evaluation_split_around_the_exchange(hostsim_engine):
    from adflow_amd.topology import BrickTopology
    try:
        hostsim_engine.set_tuning("split_eval", 2)
        hostsim_engine.set_tuning("gf_cus", 1)
        prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind, muSuthDim=1.0)
        checks.check_brick_block_res(hostsim_engine, BrickTopology(2, 1, 1, 130, 11, 24), prm, seed=23, fused_halo=True, stretch_k=2.0)
    finally:
        hostsim_engine.set_tuning("split_eval", 1)
        hostsim_engine.set_tuning("gf_cus", 0)


def test_exchange_pressure_early(hostsim_engine):
    import test_gpu_bc
    test_gpu_bc.test_exchange_pressure_early(hostsim_engine)


def test_block_res_without_intermediates(hostsim_engine):
    import test_gpu_rans
    test_gpu_rans.test_block_res_without_intermediates(hostsim_engine)


@pytest.mark.parametrize("nLayers,nranks", [(2, 2), (2, 8)])
def test_halo_pack_unpack_loopback(hostsim_engine, nLayers, nranks):
    checks.check_halo_loopback(hostsim_engine, BrickTopology(2, 2, 2, 5, 4, 3), nranks, FlowParams(equations=RANSEquations), nLayers)


@pytest.mark.parametrize("sd", [dissScalar, upwind])
def test_block_res_vs_blockette_core(hostsim_engine, sd):
    prm = FlowParams(equations=RANSEquations, spaceDiscr=sd)
    checks.check_block_res_vs_blockette(hostsim_engine, (12, 10, 9), prm, False, seed=sd, stretch_k=3.0)
    checks.check_block_res_vs_blockette(hostsim_engine, (12, 10, 9), prm, True, seed=sd, stretch_k=3.0)


# ---- adversarial states (GPU twins in test_gpu_adversarial.py)
def test_adversarial_states(hostsim_engine):
    import test_gpu_adversarial as T
    T.shock_cases(hostsim_engine, (24, 6, 5))
    T.vacuum_cases(hostsim_engine)
    T.wall_revert_case(hostsim_engine)
    T.sa_cases(hostsim_engine, (16, 6, 12))


def test_limiter_clamp_tiny_differences(hostsim_engine):
    import test_gpu_adversarial as T
    T.clamp_cases(hostsim_engine, (30, 7, 6))


_JAC_WALL = {1: -6, 2: -6, 3: -1, 4: -1, 5: -3, 6: -6}
_JAC_EULER = {1: -6, 2: -6, 3: -5, 4: -15, 5: -1, 6: -9}


@pytest.mark.parametrize("sd", [dissScalar, dissMatrix, upwind])
def test_fd_jacobian_pc_euler(hostsim_engine, sd):
    """preconditioner matrix (usePC): 7 colours, lumped dissipation, frozen sensor"""
    checks.check_fd_jacobian(hostsim_engine, (7, 6, 5), FlowParams(spaceDiscr=sd, limiter=vanAlbeda), _JAC_EULER)


def test_fd_jacobian_pc_rans(hostsim_engine):
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=vanAlbeda, orderTurb=secondOrder, acousticScaleFactor=0.5)
    checks.check_fd_jacobian(hostsim_engine, (6, 5, 4), rans, _JAC_WALL, stretch_k=2.0)
    checks.check_fd_jacobian(hostsim_engine, (6, 5, 4), rans, _JAC_WALL, frozenTurb=True, stretch_k=2.0)
    checks.check_fd_jacobian(hostsim_engine, (6, 5, 4), rans, _JAC_WALL, useTurbOnly=True, stretch_k=2.0)
    checks.check_fd_jacobian(hostsim_engine, (5, 4, 4), rans.replace(spaceDiscr=dissScalar), _JAC_WALL, viscPC=True, stretch_k=2.0)


def test_fd_jacobian_exact(hostsim_engine):
    """dR/dw (usePC = F): 13 colours (Euler) and 35 colours (viscous), the reference's own step 1e-9 at the accuracy it resolves"""
    checks.check_fd_jacobian(hostsim_engine, (6, 6, 5), FlowParams(spaceDiscr=dissScalar), _JAC_EULER, usePC=False)
    checks.check_fd_jacobian(hostsim_engine, (5, 4, 4), FlowParams(equations=NSEquations, spaceDiscr=upwind, limiter=vanAlbeda), _JAC_WALL,
                             usePC=False, stretch_k=2.0)
    checks.check_fd_jacobian(hostsim_engine, (6, 5, 4), FlowParams(spaceDiscr=upwind, limiter=minmod), _JAC_EULER, delta=1e-9, tol=1e-5)


def test_ad_jacobian(hostsim_engine):
    """forward-mode assembly: the dual-number twins of the gather kernels (kernels_ad.hip) against the reference's Tapenade routines"""
    # (the cases of tests/test_gpu_jacobian.py::test_ad_* on smaller blocks: most of the time here is the reference's own forward mode)
    import test_gpu_jacobian as tj
    from adflow_amd.params import vanAlbeda, minmod, secondOrder
    e = hostsim_engine
    checks.check_ad_jacobian(e, (8, 6, 5), FlowParams(spaceDiscr=upwind, limiter=vanAlbeda), tj.EULER)
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=vanAlbeda, orderTurb=secondOrder, acousticScaleFactor=0.5)
    checks.check_ad_jacobian(e, (8, 6, 5), rans, tj.WALL, stretch_k=2.0)
    checks.check_ad_jacobian(e, (6, 5, 4), FlowParams(spaceDiscr=dissScalar, limiter=vanAlbeda), tj.FAR, usePC=False)
    checks.check_ad_jacobian(e, (6, 5, 4), rans.replace(spaceDiscr=dissScalar), tj.WALL, usePC=False, stretch_k=2.0)
    rm = FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=minmod)
    checks.check_ad_jacobian(e, (7, 5, 4), rm, tj.WALL, frozenTurb=True, stretch_k=2.0)
    checks.check_ad_jacobian(e, (7, 5, 4), rm, tj.WALL, useTurbOnly=True, stretch_k=2.0)
    checks.check_ad_jacobian(e, (6, 5, 4), rm.replace(spaceDiscr=dissScalar), tj.WALL, viscPC=True, stretch_k=2.0)
    checks.check_ad_jacobian(e, (7, 6, 5), rm.replace(limiter=vanAlbeda, useQCR=True), tj.OPEN, usePC=False, stretch_k=2.0)
    # the preconditioner matrix of the scalar / matrix schemes by forward mode: the lumped dissipation with the frozen sensor
    # (k_inviscid_march<.., APX>) behind the thin-layer viscous march, both on dual numbers (round 6)
    checks.check_ad_jacobian(e, (6, 5, 4), rans.replace(spaceDiscr=dissScalar), tj.WALL, stretch_k=2.0)
    checks.check_ad_jacobian(e, (6, 5, 4), FlowParams(equations=NSEquations, spaceDiscr=dissMatrix, vis4=0.1), tj.WALL, stretch_k=2.0)
    checks.check_ad_jacobian(e, (6, 5, 4), FlowParams(spaceDiscr=dissScalar), tj.EULER)
    # matrix dissipation of the exact linearisation: k_inviscid_march on dual numbers behind the dual k_visc_gf (round 6)
    checks.check_ad_jacobian(e, (6, 5, 4), FlowParams(equations=NSEquations, spaceDiscr=dissMatrix, vis4=0.1), tj.WALL, usePC=False, stretch_k=2.0)


def test_pc_assemblies_on_a_level_of_several_blocks(hostsim_engine):
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
    checks.check_jacobian_several_blocks(hostsim_engine, rans, {
        1: ((6, 5, 4), {1: -6, 2: -6, 3: -1, 4: -4, 5: -3, 6: -6}, ()),
        3: ((4, 6, 5), {1: -6, 2: -15, 3: -3, 4: -6, 5: -1, 6: -1}, ())}, quick=True, stretch_k=2.0)


def test_pc_on_the_kernels_behind_the_marches(hostsim_engine):
    import test_gpu_jacobian as tj
    tj.test_pc_on_the_kernels_behind_the_marches(hostsim_engine)


def test_ad_pc_equal_states_across_a_face(hostsim_engine):
    import test_gpu_jacobian as tj
    tj.test_ad_pc_equal_states_across_a_face(hostsim_engine)


def test_pc_march_pair_of_kernels(hostsim_engine):
    """tuning pc_fused = 0: the kernels k_pc_march replaced in the preconditioner assembly (the default, 1, runs in the tests above)"""
    import test_gpu_jacobian as tj
    from adflow_amd.params import vanAlbeda
    e = hostsim_engine
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=vanAlbeda)
    try:
        e.set_tuning("pc_fused", 0)
        checks.check_fd_jacobian(e, (6, 5, 4), rans, tj.WALL, stretch_k=2.0)
        checks.check_ad_jacobian(e, (6, 5, 4), rans, tj.WALL, stretch_k=2.0)
        # ... and the marches of the default path leaving dw to k_fd_snap / k_ad_snap (jac_snap = 0; 1 runs in the tests above)
        e.set_tuning("pc_fused", 1)
        e.set_tuning("jac_snap", 0)
        checks.check_fd_jacobian(e, (6, 5, 4), rans, tj.WALL, stretch_k=2.0)
        checks.check_ad_jacobian(e, (6, 5, 4), rans, tj.WALL, stretch_k=2.0)
        # ... ny = 10: a tile whose last rows lie beyond the block
        e.set_tuning("jac_snap", 1)
        checks.check_fd_jacobian(e, (5, 10, 4), rans, tj.OPEN, stretch_k=2.0)
    finally:
        e.set_tuning("pc_fused", 1)
        e.set_tuning("jac_snap", 1)


def test_update_wall_distances_quickly(hostsim_engine):
    checks.check_wall_distance(hostsim_engine, (7, 5, 4), FlowParams(equations=RANSEquations), stretch_k=2.0)


def test_euler_wall_normal_momentum(hostsim_engine):
    import test_gpu_bc
    test_gpu_bc.test_euler_wall_normal_momentum(hostsim_engine)


def test_finalize_then_init_starts_clean(hostsim_engine):
    """adflow_gpu_finalize releases blocks, device tables, tile tables, communication patterns and side buffers: a second
    adflow_gpu_init in the same process must not see anything of the first life"""
    from adflow_amd.engine import Engine
    from hostsim.build import build
    prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
    checks.check_block_res(hostsim_engine, (9, 6, 5), prm, seed=41, stretch_k=2.0)
    checks.check_nk_residual(hostsim_engine, BrickTopology(1, 1, 1, 6, 5, 4), prm, stretch_k=2.0)
    e1 = Engine(0, _lib_path=build())          # same library instance as the fixture
    e1.close()                                 # adflow_gpu_finalize
    e2 = Engine(0, _lib_path=build())          # adflow_gpu_init again (left open: the fixture keeps using the library)
    checks.check_block_res(e2, (7, 8, 6), prm, seed=42, stretch_k=2.0)
    checks.check_nk_residual(e2, BrickTopology(1, 1, 1, 5, 6, 4), prm, stretch_k=2.0)
    hostsim_engine.blocks.clear()


def test_euler_radii_inside_the_march(hostsim_engine):
    import test_gpu_euler
    test_gpu_euler.test_euler_radii_inside_the_march(hostsim_engine)


def test_level_launches_split_over_slot_ranges(hostsim_engine):
    """gridDim.z holds (block slot, plane): with more slots than fit (lowered here by tuning max_grid_z; 65535 in production, i.e.
    ~1800 blocks of 32 planes per GPU) every level launcher calls itself on consecutive slot ranges"""
    e = hostsim_engine
    topo = BrickTopology(2, 2, 2, 5, 4, 3)
    try:
        e.set_tuning("max_grid_z", 15)            # 15 / (3 + 4) = 2 of the 8 slots per launch
        rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind, nSubIterTurb=2)
        checks.check_nk_residual(e, topo, rans, stretch_k=2.0)
        checks.check_sa_solve(e, topo, rans, stretch_k=2.0)
        checks.check_rk_smoother(e, topo, FlowParams(resAveraging=alternateResAveraging))
        checks.check_dadi_smoother(e, topo, FlowParams(equations=RANSEquations, resAveraging=noResAveraging, cfl=1.5, nSubiterations=2),
                                   stretch_k=2.0)
        checks.check_mg_cycle(e, BrickTopology(2, 2, 1, 8, 4, 4), FlowParams(), [0, 1, 0, -1], ncycles=1)
        checks.check_halo_exchange(e, topo, FlowParams(equations=RANSEquations), 2)
    finally:
        e.set_tuning("max_grid_z", 0)


def test_left_handed_block(hostsim_engine):
    import test_gpu_rans
    test_gpu_rans.test_left_handed_block(hostsim_engine)


def test_mach3_shock_default_flags(hostsim_engine):
    import test_gpu_adversarial
    test_gpu_adversarial.shock_default_flags_case(hostsim_engine, (22, 7, 6))


def test_normals_not_derived_from_the_nodes_keep_the_stored_normals(hostsim_engine):
    import test_gpu_rans
    test_gpu_rans.test_normals_not_derived_from_the_nodes_keep_the_stored_normals(hostsim_engine)


def test_random_parity_sweep(hostsim_engine):
    """a sample of tests/fuzz_parity.py: random sizes around the tile edges, random options, entry points (block_res, the blockette
    twin, approximate residual, boundary conditions, RK / D-ADI smoothers, SA solve, NK residual), all against the reference"""
    import fuzz_parity
    n, failure = fuzz_parity.sweep(hostsim_engine, 40, seed=20260926, quiet=True)
    assert failure is None, failure


def test_documented_tuning_keys_are_the_library_s(hostsim_engine):
    """DESIGN.md 8b lists the tuning keys with their defaults: every one of them is accepted (set to its default), an unknown key is
    refused -- the table and adflow_gpu_set_tuning stay in step"""
    import os
    import re
    txt = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "DESIGN.md")).read()
    i = txt.index("## 8b. Tuning keys")
    j = txt.index("## 8a.", i)
    keys = []
    for line in txt[i:j].split("\n"):
        if not line.startswith("| `"):
            continue
        cells = [c.strip() for c in line.strip("|").split("|")]
        ks = re.findall(r"`([a-z_0-9]+)`", cells[0])
        ds = [d.strip() for d in cells[1].split(",")]
        assert len(ds) == len(ks), line
        keys += list(zip(ks, ds))
    assert len(keys) >= 20, keys
    for k, d in keys:
        hostsim_engine.set_tuning(k, int(d))
    with pytest.raises(Exception):
        hostsim_engine.set_tuning("no_such_key", 1)
