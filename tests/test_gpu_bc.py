"""GPU parity (real MI355X, through the C-ABI): boundary conditions on the device
(SURVEY.md §8f "next" row 1) against the reference's own BCRoutines / turbBCRoutines:
applyAllBC_block on all six faces of a block, the smoothers, the SA solve, the multigrid
cycle and the NK residual with physical boundaries."""
import pytest

import checks
from adflow_amd.params import (FlowParams, NSEquations, RANSEquations, DADI, RungeKutta, noResAveraging, secondOrder,
                               alternateResAveraging)
from adflow_amd.topology import BrickTopology

pytestmark = pytest.mark.gpu

# BCType: -1 symm, -3 adiabatic wall, -4 isothermal wall, -5 Euler wall, -6 farfield, -7 supersonic inflow,
# -9 supersonic outflow, -15 extrap ; faces 1..6 = iMin, iMax, jMin, jMax, kMin, kMax
EULER_SPECS = [{1: -1, 2: -6, 3: -5, 4: -15, 5: -1, 6: -9}, {1: -7, 2: -6, 3: -5, 4: -5, 5: -1, 6: -1},
               {1: -6, 2: -6, 3: -6, 4: -6, 5: -5, 6: -6}]
VISC_SPECS = [{1: -6, 2: -6, 3: -1, 4: -4, 5: -3, 6: -6}, {1: -9, 2: -7, 3: -3, 4: -1, 5: -4, 6: -15}]


@pytest.mark.parametrize("spec", EULER_SPECS)
@pytest.mark.parametrize("second", [True, False])
def test_apply_all_bc_euler(engine, spec, second):
    checks.check_apply_bc(engine, (70, 9, 11), FlowParams(), spec, secondHalo=second)


@pytest.mark.parametrize("treat", [1, 2])
def test_apply_all_bc_wall_and_outflow_treatments(engine, treat):
    prm = FlowParams(eulerWallBCTreatment=treat, outflowTreatment=treat)
    checks.check_apply_bc(engine, (12, 10, 6), prm, EULER_SPECS[0])
    prm = FlowParams(equations=NSEquations, viscWallBCTreatment=treat)
    checks.check_apply_bc(engine, (12, 10, 6), prm, VISC_SPECS[0], stretch_k=2.0)


def test_euler_wall_normal_momentum(engine):
    """eulerWallBCTreatment = normalMomentum (BCRoutines.F90:1123-1234): wall pressure gradient from the normal momentum equation,
    Euler walls on all three face directions, full faces and split subfaces, a single-cell-wide subface (the clipped differences)"""
    prm = FlowParams(eulerWallBCTreatment=4)
    checks.check_apply_bc(engine, (12, 10, 6), prm, {1: -5, 2: -6, 3: -5, 4: -15, 5: -5, 6: -9})
    checks.check_apply_bc(engine, (9, 7, 5), prm, {1: -6, 2: -5, 3: -1, 4: -5, 5: -6, 6: -5}, secondHalo=False)
    checks.check_apply_bc(engine, (10, 8, 1), prm, {1: -5, 2: -6, 3: -5, 4: -6, 5: -1, 6: -1})
    checks.check_apply_bc(engine, (12, 8, 6), prm, {1: -6, 2: -6, 3: -5, 4: -5, 5: -1, 6: -5}, split={3: -6, 6: -5})


def test_exchange_pressure_early(engine):
    """normal-momentum Euler wall across a block interface: the pressure-only whalo1 in front of applyAllBC in every RK stage /
    D-ADI step (smoothers.F90:363, :674) -- the reference without it gives a different state"""
    checks.check_pressure_early_exchange(engine, (8, 6, 5), FlowParams(), wall_face=5)
    checks.check_pressure_early_exchange(engine, (6, 7, 5), FlowParams(smoother=DADI, resAveraging=noResAveraging), wall_face=3, sweeps=2)


@pytest.mark.parametrize("spec", VISC_SPECS)
def test_apply_all_bc_rans(engine, spec):
    checks.check_apply_bc(engine, (20, 7, 6), FlowParams(equations=RANSEquations), spec, stretch_k=2.0)


def test_apply_all_bc_two_dimensional_block_between_symmetry_planes(engine):
    """one cell between two symmetry planes: the reason the reference runs the 2nd-halo symmetry pass separately"""
    checks.check_apply_bc(engine, (16, 8, 1), FlowParams(), {1: -6, 2: -6, 3: -5, 4: -6, 5: -1, 6: -1})


def test_apply_all_bc_coarse_level_forces_constant_pressure_walls(engine):
    checks.check_apply_bc(engine, (12, 10, 6), FlowParams(eulerWallBCTreatment=2), EULER_SPECS[0], secondHalo=False, level=2)


def test_rk_smoother_with_bc(engine):
    checks.check_smoother_with_bc(engine, (24, 10, 6), FlowParams(smoother=RungeKutta, resAveraging=alternateResAveraging),
                                  {1: -6, 2: -6, 3: -5, 4: -6, 5: -1, 6: -1})
    checks.check_smoother_with_bc(engine, (12, 8, 6), FlowParams(equations=NSEquations, smoother=RungeKutta),
                                  {1: -6, 2: -6, 3: -1, 4: -1, 5: -3, 6: -6}, stretch_k=2.0)


def test_dadi_smoother_with_bc_rans(engine):
    prm = FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging, cfl=1.5, nSubiterations=2)
    checks.check_smoother_with_bc(engine, (12, 8, 6), prm, {1: -6, 2: -6, 3: -1, 4: -1, 5: -3, 6: -6}, stretch_k=2.0)


@pytest.mark.parametrize("order", [1, secondOrder])
def test_sa_solve_with_turbulence_bc(engine, order):
    prm = FlowParams(equations=RANSEquations, nSubIterTurb=2, orderTurb=order)
    checks.check_sa_solve_with_bc(engine, (12, 8, 6), prm, {1: -6, 2: -15, 3: -1, 4: -4, 5: -3, 6: -9}, stretch_k=2.0)
    checks.check_sa_solve_with_bc(engine, (8, 6, 5), prm, {1: -6, 2: -6, 3: -1, 4: -5, 5: -3, 6: -6}, stretch_k=2.0)


def test_mg_cycle_with_bc(engine):
    checks.check_mg_cycle(engine, BrickTopology(1, 1, 1, 16, 8, 8), FlowParams(), [0, 1, 0, -1],
                          bc_spec={1: -6, 2: -6, 3: -5, 4: -6, 5: -1, 6: -1})
    prm = FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging, cfl=1.5, nSubiterations=2,
                     nSubIterTurb=2)
    checks.check_mg_cycle(engine, BrickTopology(1, 1, 1, 8, 8, 4), prm, [0, 1, 0, -1], ncycles=2,
                          bc_spec={1: -6, 2: -6, 3: -1, 4: -1, 5: -3, 6: -6}, stretch_k=2.0)


def test_mg_cycle_three_levels_with_bc(engine):
    checks.check_mg_cycle(engine, BrickTopology(1, 1, 1, 8, 8, 8), FlowParams(equations=NSEquations),
                          [0, 1, 0, 1, 0, -1, 0, -1], ncycles=2, nlevels=3,
                          bc_spec={1: -6, 2: -9, 3: -1, 4: -15, 5: -4, 6: -6}, stretch_k=2.0)


def test_nk_residual_with_bc(engine):
    checks.check_nk_residual(engine, BrickTopology(1, 1, 1, 12, 8, 6), FlowParams(equations=RANSEquations),
                             bc_spec={1: -6, 2: -6, 3: -1, 4: -1, 5: -3, 6: -6}, stretch_k=2.0)
    checks.check_nk_residual(engine, BrickTopology(1, 1, 1, 12, 8, 6), FlowParams(),
                             bc_spec={1: -6, 2: -6, 3: -5, 4: -6, 5: -1, 6: -1})


def test_unsupported_bc_kinds_are_refused(engine):
    from adflow_amd.capi import AdflowGpuError
    from adflow_amd.synth import make_block, make_bocos
    engine.release_all()
    prm = FlowParams()
    engine.set_options(prm)
    blk = make_block(6, 5, 4, prm)
    engine.register(blk)
    faces, nv = make_bocos(blk, prm, {1: -6})
    faces[0]["bcType"] = -8          # subsonic inflow
    with pytest.raises(AdflowGpuError):
        engine.bc_register(faces, nv)


@pytest.mark.parametrize("dims", [(70, 9, 11), (16, 8, 1)])
def test_update_geometry_after_mesh_warp(engine, dims):
    """"next" row 3: volume_block + metric_block + boundaryNormals on the device"""
    checks.check_update_geometry(engine, dims, FlowParams(), {1: -1, 2: -6, 3: -5, 4: -5, 5: -1, 6: -6})
    checks.check_update_geometry(engine, dims, FlowParams(equations=NSEquations), {1: -6, 2: -6, 3: -3, 4: -6, 5: -1, 6: -1},
                                 stretch_k=2.0)


@pytest.mark.parametrize("dims", [(70, 9, 11), (16, 8, 1)])
def test_update_wall_distances_quickly(engine, dims):
    """the RANS part of the `useSpatial` branch (blockette.F90:207-209): d2Wall from the stored wall association"""
    checks.check_wall_distance(engine, dims, FlowParams(equations=RANSEquations), stretch_k=2.0)


def test_apply_all_bc_split_faces(engine):
    """block faces cut into two subfaces of different kinds (wall + farfield, symmetry + Euler wall)"""
    checks.check_apply_bc(engine, (40, 9, 6), FlowParams(), {1: -1, 2: -6, 3: -5, 4: -15, 5: -1, 6: -9}, split={3: -6, 6: -5, 1: -6})
    checks.check_apply_bc(engine, (24, 8, 6), FlowParams(equations=RANSEquations), {1: -6, 2: -6, 3: -1, 4: -4, 5: -3, 6: -6},
                          split={5: -6, 4: -3}, stretch_k=2.0)


def test_multiblock_bc(engine):
    """several blocks with different subface lists: the level-batched BC launches against the reference's block loop"""
    checks.check_multiblock_bc(engine, FlowParams(), {
        1: ((40, 9, 6), {1: -1, 2: -6, 3: -5, 4: -15, 5: -1, 6: -9}, {3: -6, 6: -5}),
        2: ((70, 7, 5), None, ()),
        3: ((12, 10, 8), {2: -6, 3: -5, 6: -7}, ()),
        4: ((9, 8, 8), {1: -5, 2: -5, 3: -5, 4: -5, 5: -5, 6: -5}, {1: -6})})
    checks.check_multiblock_bc(engine, FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging), {
        1: ((24, 8, 6), {1: -6, 2: -6, 3: -1, 4: -4, 5: -3, 6: -6}, {5: -6, 4: -3}),
        2: ((70, 6, 6), {3: -3, 4: -6}, ()),
        3: ((10, 12, 8), {1: -6, 2: -9, 3: -3, 4: -6, 5: -1, 6: -1}, ())}, stretch_k=2.0)


def test_apply_all_bc_subsonic_and_polar(engine):
    """symmPolar, subsonic inflow (total conditions on min faces, mass flow on max faces, hScalingInlet), subsonic
    outflow / outflow bleed, with the turbulence inflow / outflow treatment for RANS"""
    lo = dict(Mach=0.3)
    spec = {1: -8, 2: -10, 3: -2, 4: -8, 5: -12, 6: -6}
    for second in (True, False):
        checks.check_apply_bc(engine, (70, 9, 7), FlowParams(**lo), spec, secondHalo=second)
    checks.check_apply_bc(engine, (20, 12, 8), FlowParams(hScalingInlet=True, **lo), {1: -10, 2: -8, 3: -8, 4: -2, 5: -1, 6: -5})
    rans = FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging, **lo)
    checks.check_multiblock_bc(engine, rans, {
        1: ((24, 8, 6), {1: -8, 2: -10, 3: -3, 4: -6, 5: -2, 6: -7}, ()),
        2: ((70, 6, 8), {1: -12, 2: -8, 3: -3, 4: -6}, {3: -6})}, stretch_k=2.0)


def test_coordinate_halos(engine):
    """"next" row 3: xhalo_block + exchangeCoor (node pattern) + metrics after a mesh warp"""
    checks.check_coordinate_halos_brick(engine, BrickTopology(2, 2, 1, 20, 9, 8), FlowParams())
    checks.check_coordinate_halos_brick(engine, BrickTopology(1, 1, 2, 70, 6, 4), FlowParams(equations=NSEquations), stretch_k=2.0)
    checks.check_xhalo_symmetry(engine, (70, 9, 8), FlowParams(), {1: -1, 2: -6, 3: -5, 4: -1, 5: -1, 6: -6})
    checks.check_xhalo_symmetry(engine, (24, 10, 8), FlowParams(), {1: -1, 2: -1, 3: -1, 4: -6, 5: -6, 6: -1}, split={3: -6, 6: -5})
    checks.check_coarse_level_geometry(engine, BrickTopology(2, 1, 1, 24, 12, 8), FlowParams())
    # odd cell counts: the coarse nodes are the nodes createCoarseBlocks kept (1, 3, .., il - 1, il), not every second one
    checks.check_coarse_level_geometry(engine, BrickTopology(2, 1, 1, 17, 13, 9), FlowParams())


def test_bc_on_large_faces(engine):
    """faces large enough to need several workgroups per launch: every kind, split faces, turbulence + mean flow together inside
    blocketteRes, smoothers, SA solve, multigrid with boundary subfaces"""
    for spec in EULER_SPECS:
        checks.check_apply_bc(engine, (70, 19, 11), FlowParams(), spec)
    checks.check_apply_bc(engine, (30, 22, 14), FlowParams(), EULER_SPECS[0], secondHalo=False)
    for spec in VISC_SPECS:
        checks.check_apply_bc(engine, (26, 20, 12), FlowParams(equations=RANSEquations), spec, stretch_k=2.0)
    checks.check_apply_bc(engine, (24, 18, 12), FlowParams(), {1: -1, 2: -1, 3: -1, 4: -6, 5: -6, 6: -1}, split={3: -6, 6: -5})
    checks.check_apply_bc(engine, (14, 11, 9), FlowParams(outflowTreatment=2, hScalingInlet=True), {1: -8, 2: -8, 3: -10, 4: -12, 5: -2, 6: -6})
    rans = FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging, cfl=1.5, nSubiterations=2, nSubIterTurb=2)
    wall = {1: -6, 2: -6, 3: -1, 4: -1, 5: -3, 6: -6}
    checks.check_smoother_with_bc(engine, (24, 16, 12), rans, wall, stretch_k=2.0)
    checks.check_sa_solve_with_bc(engine, (20, 14, 10), rans.replace(orderTurb=secondOrder), {1: -6, 2: -15, 3: -1, 4: -4, 5: -3, 6: -9}, stretch_k=2.0)
    checks.check_blockette_res_with_bc(engine, BrickTopology(2, 1, 2, 24, 16, 12, periodic=(False, False, False)),
                                       FlowParams(equations=RANSEquations), {1: -6, 2: -6, 3: -1, 4: -6, 5: -3, 6: -6}, stretch_k=2.0)
    checks.check_mg_cycle(engine, BrickTopology(1, 1, 1, 32, 24, 16), FlowParams(), [0, 1, 0, -1], bc_spec={1: -6, 2: -6, 3: -5, 4: -6, 5: -1, 6: -1})
