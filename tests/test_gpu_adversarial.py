"""GPU parity on ADVERSARIAL states (tests/adversarial.py): each test first asserts, by a host-side recount on the arrays the
reference runs on, that the branch it is about really fires -- dissipation clips and the Roe entropy fix at a Mach-3 shock,
limiter cut-off / epsLim clamp, the density / pressure floor of the Runge-Kutta stage update, the negative-pressure revert of
the linear wall extrapolation, the Spalart-Allmaras limiters -- and then compares with the reference's own Fortran."""
import numpy as np
import pytest

import adversarial
import checks
from adflow_amd.params import (FlowParams, dissScalar, dissMatrix, upwind, vanAlbeda, minmod, noLimiter, NSEquations, RANSEquations,
                               secondOrder, noResAveraging)
from adflow_amd.topology import BrickTopology

pytestmark = pytest.mark.gpu


def shock_cases(engine, dims):
    for eq in (1, RANSEquations):
        for sd, lim in ((dissScalar, vanAlbeda), (dissMatrix, vanAlbeda), (upwind, vanAlbeda), (upwind, minmod), (upwind, noLimiter)):
            prm = FlowParams(equations=eq, spaceDiscr=sd, limiter=lim, vis4=0.05 if sd == dissMatrix else 0.0156)   # 0.05 < vis2 / 4: the dis4 clip can fire
            blk = adversarial.shock_block(dims, prm, seed=sd + lim, stretch_k=2.0 if eq != 1 else 1.0)
            n = adversarial.count_shock_branches(blk, prm)
            assert min(n.values()) > 0, n
            checks.check_block_res(engine, dims, prm, blk=blk)


def test_mach3_shock_dissipation_clips_entropy_fix_limiter_extremes(engine):
    shock_cases(engine, (70, 9, 6))


def clamp_cases(engine, dims):
    """differences inside the epsLim clamp of the limiters: k_roe_march takes its symmetric van Albada form only where no difference
    of a reconstruction lies in (0, epsLim) and the clamped form of the reference elsewhere"""
    for eq in (1, RANSEquations):
        for lim in (vanAlbeda, minmod):
            prm = FlowParams(equations=eq, spaceDiscr=upwind, limiter=lim)
            blk = adversarial.clamp_block(dims, prm, seed=40 + lim, stretch_k=2.0 if eq != 1 else 1.0)
            assert adversarial.count_clamped_differences(blk) > 1000
            checks.check_block_res(engine, dims, prm, blk=blk)


def test_limiter_clamp_tiny_differences(engine):
    clamp_cases(engine, (70, 9, 6))


def shock_default_flags_case(engine, dims):
    """the same Mach-3 shock through blocketteRes with its DEFAULT flags: for Euler + scalar JST the march forms the spectral radii
    itself (fast_powa, rcp / rsq forms) on states whose radii span orders of magnitude; RANS takes the marching kernels"""
    from oracle import ref
    from util import owned, rel_err, TOL
    for eq, sd in ((1, dissScalar), (3, upwind), (3, dissScalar)):
        prm = FlowParams(equations=eq, spaceDiscr=sd)
        blk = adversarial.shock_block(dims, prm, seed=90 + sd, stretch_k=2.0 if eq != 1 else 1.0)
        assert min(adversarial.count_shock_branches(blk, prm).values()) > 0
        engine.release_all()
        r = checks.ref_bind(blk, prm)
        ref.block_res_core(False, True, eq == 3)
        engine.set_options(prm)
        engine.register(blk)
        engine.blocketteRes(1, False, True, eq == 3)
        dw = engine.download_residual()
        e = rel_err(owned(blk, dw), owned(blk, r["dw"]))
        assert e <= TOL, (eq, sd, e)


def test_mach3_shock_default_flags(engine):
    shock_default_flags_case(engine, (70, 9, 6))


def vacuum_cases(engine):
    n = checks.check_vacuum_smoother(engine, BrickTopology(1, 1, 1, 12, 8, 6), FlowParams(resAveraging=noResAveraging))
    assert n > 0, "the reference clipped no cell: the state is not adversarial enough"
    n = checks.check_vacuum_smoother(engine, BrickTopology(2, 1, 1, 12, 9, 9), FlowParams(equations=NSEquations, resAveraging=noResAveraging),
                                     frac=5e-5, stretch_k=2.0)
    assert n > 0


def test_near_vacuum_stage_update_clipping(engine):
    vacuum_cases(engine)


def wall_revert_case(engine):
    hits = []

    def mutate(blk):
        # third cell off the kMin wall three times the pressure of the second: 2 p2 - p3 < 0 -> the halo reverts to p2
        p = blk["p"]
        p[2:-2:2, 2:-2, 3] = 3.0 * p[2:-2:2, 2:-2, 2]
        hits.append(int((2.0 * p[2:-2, 2:-2, 2] - p[2:-2, 2:-2, 3] <= 0).sum()))
    prm = FlowParams(equations=NSEquations, viscWallBCTreatment=2)
    checks.check_apply_bc(engine, (10, 8, 6), prm, {1: -6, 2: -6, 3: -6, 4: -6, 5: -3, 6: -6}, mutate=mutate, stretch_k=2.0)
    assert hits and hits[0] > 0


def test_wall_linear_pressure_extrapolation_reverts_when_negative(engine):
    wall_revert_case(engine)


def sa_cases(engine, dims):
    for order in (1, secondOrder):
        prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind, orderTurb=order)
        blk = adversarial.sa_extremes(dims, prm, seed=4 + order, stretch_k=2.0)
        n = adversarial.count_sa_branches(blk, prm)
        assert min(n.values()) > 0, n
        checks.check_block_res(engine, dims, prm, blk=blk)
    prm = FlowParams(equations=RANSEquations, useft2SA=False, useRotationSA=True)
    blk = adversarial.sa_extremes(dims, prm, seed=9, stretch_k=2.0)
    checks.check_block_res(engine, dims, prm, blk=blk)


def test_spalart_allmaras_limiters(engine):
    sa_cases(engine, (24, 8, 16))


def test_spalart_allmaras_negative_working_variable(engine):
    """nuTilde < 0 in a pocket close to the wall (a transient of a diverging outer iteration): sst is floored at 1e-10, rr is only
    bounded from above (sa.F90), so rr ~ -1e9 and the argument of the sixth root in fw drops to ~1e-60 and below -- outside the
    range of a single-precision seed (round-2 advisor finding).  The residual must stay finite and equal to the reference's."""
    import numpy as np
    from adflow_amd.synth import make_block
    prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
    dims = (24, 8, 16)
    blk = make_block(*dims, prm, seed=19, stretch_k=2.0)
    w = blk["w"]
    nu = blk["rlv"] / w[..., 0]
    w[6:14, :, 3:7, 5] = -50.0 * nu[6:14, :, 3:7]
    w[15:18, :, 3:6, 5] = -1e6 * nu[15:18, :, 3:6]
    blk["d2Wall"][:, :, :6] = 1e-6
    r = checks.check_block_res(engine, dims, prm, blk=blk)


def test_random_parity_sweep(engine):
    """tests/fuzz_parity.py on the GPU: 300 random cases (sizes around the tile edges, options, entry points) against the reference"""
    import fuzz_parity
    n, failure = fuzz_parity.sweep(engine, 300, seed=20260926, quiet=True)
    assert failure is None, failure
