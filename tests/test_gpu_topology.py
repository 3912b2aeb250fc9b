"""GPU parity (real MI355X, through the C-ABI): 1-to-1 interfaces that are NOT axis-aligned twins (round-4 verdict, missing 3).
The reference's subfaces carry a transformation between the index systems of the two blocks (modules/block.F90:271-309
l1, l2, l3) and join blocks of different dimensions; its preprocessing turns them into the index lists the hot path consumes.
adflow_amd.topology.ell_topology builds those lists for four blocks -- 24x16x8, 16x12x8 joined through its jMin face with its i
running AGAINST the first block's j, 16x6x24 joined through a cyclic permutation of the indices, a LEFT-handed 6x16x12 block; blocks meet along edges
too (indirect halos) -- and every entry point that consumes lists or assumes something about interfaces runs on it against the
reference's own routines: whalo1 / whalo2, the pack / unpack leg, the RCCL leg to the own rank, exchangeCoor + xhalo + metrics,
blocketteRes with boundary conditions (also split around the exchange), exchangePressureEarly, smoothers, SA solve, matvec glue
and multigrid cycles with and without boundary subfaces."""
import pytest

import checks
from adflow_amd.params import FlowParams, RANSEquations, DADI, noResAveraging, upwind, dissMatrix
from adflow_amd.topology import ell_topology

pytestmark = pytest.mark.gpu

RANS = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
WALLS = {1: -6, 2: -6, 3: -1, 4: -6, 5: -3, 6: -6}        # by outward lattice direction: symmetry at -y, viscous wall at -z, farfield


@pytest.mark.parametrize("nLayers", [1, 2])
def test_halo_exchange_rotated_interfaces(engine, nLayers):
    checks.check_halo_exchange(engine, ell_topology(), RANS, nLayers)
    checks.check_halo_loopback(engine, ell_topology(), 2, FlowParams(), nLayers)
    checks.check_halo_loopback(engine, ell_topology(), 3, RANS, nLayers)


def test_halo_exchange_rotated_interfaces_rccl_self(engine):
    """every interface of the three blocks as a message to the own rank: k_halo_pack -> ncclSend / ncclRecv -> k_halo_unpack"""
    engine.comm_init_single()
    try:
        engine.set_tuning("comm_self", 1)
        for nLayers in (1, 2):
            checks.check_halo_exchange(engine, ell_topology(), RANS, nLayers)
        checks.check_rk_smoother(engine, ell_topology(), FlowParams())
        checks.check_brick_block_res(engine, ell_topology(2), RANS, seed=29, fused_halo=True)
    finally:
        engine.set_tuning("comm_self", 0)


def test_coordinate_halos_rotated_interfaces(engine):
    checks.check_coordinate_halos_brick(engine, ell_topology(), FlowParams())
    checks.check_coarse_level_geometry(engine, ell_topology(), FlowParams())


@pytest.mark.parametrize("split_eval", [None, 2])
def test_blockette_res_rotated_interfaces(engine, split_eval):
    """the whole blocketteRes (derived values, BCs, whalo2, core, wall stress) with 13 boundary subfaces around the three blocks;
    split_eval = 2: the halo-free tiles between departure and arrival of the exchange"""
    n = checks.check_blockette_res_with_bc(engine, ell_topology(stretch_z=2.0), RANS, WALLS, split_eval=split_eval)
    assert n == 2                                   # the viscous wall lies under blocks A and B only
    checks.check_blockette_res_with_bc(engine, ell_topology(2, stretch_z=2.0), RANS.replace(spaceDiscr=dissMatrix, vis4=0.1, useQCR=True), WALLS,
                                       split_eval=split_eval, seed=23)
    checks.check_brick_block_res(engine, ell_topology(), RANS, fused_halo=True)


def test_exchange_pressure_early_rotated_interface(engine):
    """an inviscid wall (normal-momentum extrapolation) under blocks A and B: the pressure derivative along the wall crosses the
    interface where A's i direction continues as B's j"""
    spec = {1: -6, 2: -6, 3: -6, 4: -6, 5: -5, 6: -6}
    moved = checks.check_pressure_early_exchange(engine, None, FlowParams(), topo=ell_topology(), lattice_spec=spec)
    assert moved > 1e-6
    checks.check_pressure_early_exchange(engine, None, FlowParams(smoother=DADI, resAveraging=noResAveraging, cfl=1.5), topo=ell_topology(),
                                         lattice_spec=spec)


def test_smoothers_rotated_interfaces(engine):
    checks.check_rk_smoother(engine, ell_topology(), FlowParams(), nsweeps=2)
    checks.check_dadi_smoother(engine, ell_topology(), FlowParams(resAveraging=noResAveraging, cfl=1.5))
    checks.check_sa_solve(engine, ell_topology(stretch_z=2.0), FlowParams(equations=RANSEquations, nSubIterTurb=2))
    checks.check_nk_residual(engine, ell_topology(), RANS)


def test_mg_cycle_rotated_interfaces(engine):
    # (without boundary subfaces the outer halos of the coarse levels would never be written)
    checks.check_mg_cycle(engine, ell_topology(), FlowParams(), [0, 1, 0, -1], brick_spec={1: -6, 2: -6, 3: -1, 4: -6, 5: -5, 6: -6})
    rans = FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging, cfl=1.5, nSubiterations=2, nSubIterTurb=2)
    checks.check_mg_cycle(engine, ell_topology(stretch_z=2.0), rans, [0, 1, 0, -1], ncycles=1, brick_spec=WALLS)
