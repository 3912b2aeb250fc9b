"""GPU parity (real MI355X, through the C-ABI): multigrid restriction /
prolongation and full cycles against the reference's own transferToCoarseGrid,
transferToFineGrid and executeMGCycle (src/solver/multiGrid.F90) on two-level
periodic bricks (BASELINE config 2: multiblock Euler, RK multigrid)."""
import pytest

import checks
from adflow_amd.params import FlowParams, NSEquations, DADI, noResAveraging, upwind
from adflow_amd.topology import BrickTopology

pytestmark = pytest.mark.gpu


def test_mg_transfer_operators(engine):
    checks.check_mg_transfer(engine, BrickTopology(2, 1, 1, 16, 12, 8), FlowParams(resAveraging=noResAveraging))


@pytest.mark.parametrize("cycling", [[0, 1, 0, -1], [0, 1, 0, 0, -1, 0]])
def test_mg_cycle_euler_rk_tutorial_wing_size(engine, cycling):
    # 6 blocks x (16x14x8): BASELINE config 2 parity size, default alternate residual averaging
    checks.check_mg_cycle(engine, BrickTopology(3, 2, 1, 16, 14, 8), FlowParams(), cycling)


def test_mg_cycle_laminar(engine):
    checks.check_mg_cycle(engine, BrickTopology(2, 1, 1, 12, 8, 8), FlowParams(equations=NSEquations, resAveraging=noResAveraging),
                          [0, 1, 0, -1], stretch_k=2.0)


def test_mg_cycle_dadi(engine):
    checks.check_mg_cycle(engine, BrickTopology(2, 1, 1, 12, 8, 8), FlowParams(smoother=DADI, resAveraging=noResAveraging, cfl=1.5),
                          [0, 1, 0, -1])


def test_mg_cycle_upwind(engine):
    prm = FlowParams(spaceDiscr=upwind, spaceDiscrCoarse=upwind, resAveraging=noResAveraging)
    checks.check_mg_cycle(engine, BrickTopology(2, 1, 1, 12, 8, 8), prm, [0, 1, 0, -1])


def test_mg_cycle_rans_dadi_with_sa_solve(engine):
    # BASELINE config 3: RANS-SA, D-ADI smoother, turbSolveDDADI closing the cycle
    from adflow_amd.params import RANSEquations
    prm = FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging, cfl=1.5, nSubiterations=3,
                     nSubIterTurb=3)
    checks.check_mg_cycle(engine, BrickTopology(2, 1, 1, 12, 8, 8), prm, [0], ncycles=2, stretch_k=2.5)


def test_mg_cycle_matrix_dissipation_on_coarse_level(engine):
    from adflow_amd.params import dissMatrix
    prm = FlowParams(spaceDiscr=dissMatrix, spaceDiscrCoarse=dissMatrix, vis4=0.1, resAveraging=noResAveraging)
    checks.check_mg_cycle(engine, BrickTopology(2, 1, 1, 12, 8, 8), prm, [0, 1, 0, -1])


def test_mg_cycle_three_levels_v(engine):
    """3-level V cycle (mgStartlevel 1, cycle strategy "3v"-like)"""
    checks.check_mg_cycle(engine, BrickTopology(2, 1, 1, 16, 8, 8), FlowParams(), [0, 1, 0, 1, 0, -1, 0, -1], nlevels=3)


def test_mg_cycle_three_levels_w(engine):
    """3-level W cycle: the coarsest level is visited twice"""
    checks.check_mg_cycle(engine, BrickTopology(1, 1, 1, 16, 8, 8), FlowParams(),
                          [0, 1, 0, 1, 0, -1, 0, 1, 0, -1, 0, -1, 0], nlevels=3)


def test_mg_3w_cycle_larger_blocks(engine):
    """3-level W cycle on two 64 x 48 x 32 blocks: several tiles and k chunks on the fine level, 16 x 12 x 8 blocks on the coarsest"""
    from adflow_amd.topology import BrickTopology
    checks.check_mg_cycle(engine, BrickTopology(2, 1, 1, 64, 48, 32), FlowParams(), [0, 1, 0, 1, 0, -1, 0, 1, 0, -1, 0, -1],
                          ncycles=1, nlevels=3)


def test_mg_cycle_graph_replay(engine):
    """adflow_gpu_mg_cycle captures the second identical cycle into a hipGraph and replays it from the third on (tuning mg_graph):
    four cycles in a row against the reference, RK W cycle with residual averaging and RANS D-ADI + SA solve; then the same without
    the graph, and a state upload between the cycles (other block flags than the captured ones: that call runs directly)"""
    from adflow_amd.params import RANSEquations
    rans = FlowParams(equations=RANSEquations, smoother=DADI, resAveraging=noResAveraging, cfl=1.5, nSubiterations=2, nSubIterTurb=2)
    try:
        for g in (1, 0):
            engine.set_tuning("mg_graph", g)
            checks.check_mg_cycle(engine, BrickTopology(2, 1, 1, 16, 8, 8), FlowParams(), [0, 1, 0, 1, 0, -1, 0, 1, 0, -1, 0, -1], ncycles=4,
                                  nlevels=3)
            checks.check_mg_cycle(engine, BrickTopology(2, 1, 1, 12, 8, 8), rans, [0, 1, 0, -1], ncycles=4, stretch_k=2.5)
    finally:
        engine.set_tuning("mg_graph", 1)
