"""GPU parity (real MI355X, through the C-ABI): halo exchange, Runge-Kutta and
D-ADI smoother sweeps against the reference's own shell routines
(haloExchange.F90 whalo1/whalo2, smoothers.F90 RungeKuttaSmoother/DADISmoother)
on periodic bricks of blocks (no physical boundaries)."""
import pytest

import checks
from adflow_amd.params import (FlowParams, NSEquations, RANSEquations, noResAveraging, alternateResAveraging,
                               alwaysResAveraging, dissMatrix, upwind)
from adflow_amd.topology import BrickTopology

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nLayers", [1, 2])
def test_halo_exchange_same_gpu(engine, nLayers):
    checks.check_halo_exchange(engine, BrickTopology(2, 2, 1, 8, 6, 5), FlowParams(equations=RANSEquations), nLayers)


@pytest.mark.parametrize("nLayers,nranks", [(2, 2), (1, 2), (2, 8)])
def test_halo_pack_unpack_loopback(engine, nLayers, nranks):
    """a18: the pack / unpack kernels of the inter-GPU path on the real device (2 and 8 virtual ranks of a 2x2x2 brick,
    every rank with face, edge and corner peers) against the reference's whalo1 / whalo2 on the undivided brick"""
    checks.check_halo_loopback(engine, BrickTopology(2, 2, 2, 8, 6, 5), nranks, FlowParams(equations=RANSEquations), nLayers)


@pytest.mark.parametrize("nLayers", [2, 1])
def test_halo_exchange_rccl_self_loopback(engine, nLayers):
    """the RCCL leg of whalo1 / whalo2 (api.hip comm_exchange_enqueue: k_halo_pack -> ncclGroupStart .. ncclSend / ncclRecv ..
    ncclGroupEnd on the communication queue -> k_halo_unpack; replaces mpi_isend / mpi_irecv / waitany of
    haloExchange.F90:553-719) EXECUTED on one GPU: tuning comm_self routes every same-process interface through a message to
    the own rank.  Compared with the reference's whalo1 / whalo2, then one RK sweep with the exchange between the stages."""
    engine.comm_init_single()
    assert engine.comm_info() == (0, 1, 1, 0)      # what the RCCL communicator itself reports: one rank, this one
    try:
        engine.set_tuning("comm_self", 1)
        checks.check_halo_exchange(engine, BrickTopology(2, 2, 2, 8, 6, 5), FlowParams(equations=RANSEquations), nLayers)
        checks.check_halo_exchange(engine, BrickTopology(1, 1, 1, 7, 5, 3), FlowParams(), nLayers)
        if nLayers == 2:
            checks.check_rk_smoother(engine, BrickTopology(2, 1, 2, 12, 8, 6), FlowParams())
            # whalo2 inside blocketteRes: the halo-free tiles run while the messages to the own rank are in flight (split_eval = 1)
            engine.set_tuning("gf_cus", 2)          # small rounds: several k chunks per column, interior ones among them
            checks.check_brick_block_res(engine, BrickTopology(2, 1, 1, 130, 14, 40), FlowParams(equations=RANSEquations, spaceDiscr=upwind),
                                         seed=29, fused_halo=True, stretch_k=2.0)
            engine.set_tuning("gf_cus", 0)
    finally:
        engine.set_tuning("comm_self", 0)


def test_halo_exchange_self_periodic_single_block(engine):
    checks.check_halo_exchange(engine, BrickTopology(1, 1, 1, 7, 5, 3), FlowParams(), 2)


@pytest.mark.parametrize("resavg", [noResAveraging, alternateResAveraging, alwaysResAveraging])
def test_rk_smoother_euler(engine, resavg):
    checks.check_rk_smoother(engine, BrickTopology(2, 1, 1, 16, 12, 8), FlowParams(resAveraging=resavg), nsweeps=2)


def test_res_averaging_line_bundles(engine):
    """k_ra_line (lines resident in LDS): several bundles per direction with a ragged last one, blanked cells, a 2 x 2 x 2 brick at a
    third of the bench size, lines too long for the LDS buffer (two-pass kernels), one-cell directions"""
    prm = FlowParams(resAveraging=alwaysResAveraging)
    checks.check_rk_smoother(engine, BrickTopology(1, 1, 1, 70, 37, 19), prm, holes=0.05)
    checks.check_rk_smoother(engine, BrickTopology(2, 2, 2, 48, 40, 36), FlowParams(resAveraging=alternateResAveraging))
    checks.check_rk_smoother(engine, BrickTopology(1, 1, 1, 3, 2, 19), prm)
    checks.check_rk_smoother(engine, BrickTopology(1, 1, 1, 1340, 2, 1), prm)
    checks.check_rk_smoother(engine, BrickTopology(1, 1, 1, 2, 1040, 3), prm)
    checks.check_rk_smoother(engine, BrickTopology(1, 1, 1, 3, 2, 1040), prm)
    # the i direction by cyclic reduction (k_ra_i_pcr, lines up to 256 cells: 1 .. 4 wavefronts per line); ra_pcr = 0: the
    # LDS-resident lines on the same cases
    for nx in (130, 200, 250):
        checks.check_rk_smoother(engine, BrickTopology(1, 1, 1, nx, 5, 4), prm, holes=0.05)
    try:
        engine.set_tuning("ra_pcr", 0)
        checks.check_rk_smoother(engine, BrickTopology(1, 1, 1, 70, 37, 19), prm, holes=0.05)
        checks.check_rk_smoother(engine, BrickTopology(1, 1, 1, 130, 5, 4), prm, holes=0.05)
    finally:
        engine.set_tuning("ra_pcr", 1)


def test_rk_smoother_multiblock_tutorial_wing_size(engine):
    # BASELINE config 2 parity size: 6 blocks, ~12 096 cells
    checks.check_rk_smoother(engine, BrickTopology(3, 2, 1, 16, 14, 9), FlowParams(), nsweeps=2)


@pytest.mark.parametrize("sd", [dissMatrix, upwind])
def test_rk_smoother_other_schemes(engine, sd):
    checks.check_rk_smoother(engine, BrickTopology(2, 1, 1, 10, 8, 6), FlowParams(spaceDiscr=sd, resAveraging=noResAveraging))


def test_rk_smoother_rans(engine):
    checks.check_rk_smoother(engine, BrickTopology(2, 1, 1, 12, 10, 8),
                             FlowParams(equations=RANSEquations, resAveraging=noResAveraging), stretch_k=2.0)


def test_dadi_smoother_euler(engine):
    checks.check_dadi_smoother(engine, BrickTopology(2, 1, 1, 12, 10, 8), FlowParams(resAveraging=noResAveraging, cfl=1.5),
                               nsweeps=2)


def test_dadi_smoother_rans_tutorial_wing_config(engine):
    # BASELINE config 3: RANS-SA, D-ADI, nSubiter = 3, cfl 1.5, no residual averaging
    prm = FlowParams(equations=RANSEquations, resAveraging=noResAveraging, cfl=1.5, nSubiterations=3)
    checks.check_dadi_smoother(engine, BrickTopology(2, 2, 1, 12, 10, 8), prm, stretch_k=2.5)


def test_dadi_and_sa_solve_north_star_size_block(engine):
    """the line-solve kernels at the block size of the north-star mesh (160 x 128 x 64: 160 is no multiple of the 64-line
    workgroups nor of the 8-cell chunks of the tiled i sweeps), D-ADI with two sub-iterations and the SA DDADI solve, against
    the reference's DADISmoother / sa_block"""
    prm = FlowParams(equations=RANSEquations, resAveraging=noResAveraging, cfl=1.5, nSubiterations=2, nSubIterTurb=1)
    checks.check_dadi_smoother(engine, BrickTopology(1, 1, 1, 160, 128, 64), prm, stretch_k=3.0)
    checks.check_sa_solve(engine, BrickTopology(1, 1, 1, 160, 128, 64), prm, stretch_k=3.0)


def test_rk_smoother_with_residual_averaging_config_1_size_block(engine):
    """RK5 + alternate residual averaging on a 128^3 Euler block (the block size of BASELINE config 1 in bench.py): four k chunks
    and three i tiles of the march, 128-cell lines of the averaging sweeps"""
    checks.check_rk_smoother(engine, BrickTopology(1, 1, 1, 128, 128, 128), FlowParams(resAveraging=alternateResAveraging))


def test_dadi_i_direction_by_cyclic_reduction(engine):
    """k_dadi_i_pcr: i lines of 70 / 130 / 200 cells (workgroups of 2 / 3 / 4 wavefronts, reduction steps across the wavefronts), a
    9-cell line, one-cell lines; tuning dadi_pcr = 0 (rows + tiled Thomas: the form for lines beyond 256 cells) on the same case"""
    prm = FlowParams(equations=RANSEquations, resAveraging=noResAveraging, cfl=1.5, nSubiterations=2)
    for nx, ny, nz in ((70, 5, 4), (130, 9, 3), (200, 3, 5), (9, 8, 7)):
        checks.check_dadi_smoother(engine, BrickTopology(1, 1, 1, nx, ny, nz), prm, stretch_k=2.0)
    checks.check_dadi_smoother(engine, BrickTopology(1, 2, 1, 1, 6, 5), FlowParams(resAveraging=noResAveraging, cfl=1.5, nSubiterations=2))
    # the same scheme in the i direction of the SA DDADI solve (k_sa_i_pcr)
    for nx, ny, nz in ((70, 5, 4), (130, 9, 3), (200, 3, 5), (9, 8, 7)):
        checks.check_sa_solve(engine, BrickTopology(1, 1, 1, nx, ny, nz), prm.replace(nSubIterTurb=2), stretch_k=2.0)
    try:
        engine.set_tuning("dadi_pcr", 0)
        checks.check_dadi_smoother(engine, BrickTopology(1, 1, 1, 130, 9, 3), prm, stretch_k=2.0)
        engine.set_tuning("dadi_pcr", 1)
        # the update as its own pass
        engine.set_tuning("dadi_upd", 0)
        checks.check_dadi_smoother(engine, BrickTopology(1, 2, 1, 20, 9, 7), prm, stretch_k=2.0)
    finally:
        engine.set_tuning("dadi_pcr", 1)
        engine.set_tuning("dadi_upd", 1)


def test_dadi_degenerate_lines(engine):
    prm = FlowParams(equations=NSEquations, resAveraging=noResAveraging, cfl=1.5, nSubiterations=2)
    checks.check_dadi_smoother(engine, BrickTopology(1, 1, 2, 5, 1, 4), prm, stretch_k=2.0)


@pytest.mark.parametrize("nsub,order", [(1, 1), (3, 2)])
def test_sa_ddadi_solve(engine, nsub, order):
    prm = FlowParams(equations=RANSEquations, nSubIterTurb=nsub, orderTurb=order)
    checks.check_sa_solve(engine, BrickTopology(2, 2, 1, 12, 10, 8), prm, stretch_k=2.5)


def test_smoothers_with_blanked_cells(engine):
    topo = BrickTopology(2, 1, 1, 12, 10, 8)
    checks.check_rk_smoother(engine, topo, FlowParams(), holes=0.1)
    checks.check_dadi_smoother(engine, topo, FlowParams(resAveraging=noResAveraging, cfl=1.5), holes=0.1)
    checks.check_sa_solve(engine, topo, FlowParams(equations=RANSEquations, nSubIterTurb=2), holes=0.1, stretch_k=2.0)


def test_low_speed_preconditioner(engine):
    """residual_block's 5x5 low-Mach transform (residuals.F90:172-331) and the 0.8 RK step factor (smoothers.F90:202);
    blocketteRes does not apply it (blockette.F90:755-852)"""
    from adflow_amd.params import DADI, upwind
    lo = dict(lowSpeedPreconditioner=True, Mach=0.15)
    checks.check_rk_residual_sequence(engine, (70, 9, 7), FlowParams(**lo))
    checks.check_rk_residual_sequence(engine, (24, 10, 8), FlowParams(equations=RANSEquations, spaceDiscr=upwind, **lo), stretch_k=2.0)
    checks.check_rk_smoother(engine, BrickTopology(2, 1, 1, 20, 9, 8), FlowParams(resAveraging=alternateResAveraging, **lo))
    checks.check_dadi_smoother(engine, BrickTopology(1, 2, 1, 16, 8, 8),
                               FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging, **lo), stretch_k=2.0)
    checks.check_block_res(engine, (20, 10, 8), FlowParams(**lo), seed=5)


def test_periodic_halos(engine):
    """a18: periodic transformations of the halo exchange (velocities) and of exchangeCoor (coordinates)"""
    checks.check_periodic_halos(engine, BrickTopology(2, 1, 1, 20, 9, 8), FlowParams())
    checks.check_periodic_halos(engine, BrickTopology(1, 2, 1, 70, 6, 4), FlowParams(equations=RANSEquations), stretch_k=2.0)


def test_level_launches_split_over_slot_ranges(engine):
    import test_hostsim_kernels
    test_hostsim_kernels.test_level_launches_split_over_slot_ranges(engine)
