"""Parity checks shared by the GPU tests (-m gpu, real MI355X through the C-ABI)
and the CPU-only kernel-logic tests (tests/hostsim emulator).  Every check
compares against the reference's OWN Fortran (oracle/_ref)."""
import itertools

import numpy as np

from adflow_amd import capi
from adflow_amd.params import (FlowParams, EulerEquations, NSEquations, RANSEquations, dissScalar, dissMatrix, upwind,
                               noLimiter, vanAlbeda, minmod, RungeKutta, DADI)
from adflow_amd.synth import make_block
from util import TOL, rel_err, owned

def new_level(engine):
    """Entry points act on ALL blocks of a level: start every check from an
    empty registry and use the fine level."""
    engine.release_all()
    return 1


def ref_bind(blk, prm):
    from oracle import ref
    b = blk.copy()
    ref.bind_block(b, prm.replace(currentLevel=1, groundLevel=1))
    return b


def assert_dw(blk, dw_gpu, dw_ref, nvar=5, tol=TOL, what="dw"):
    for l in range(nvar):
        e = rel_err(owned(blk, dw_gpu[..., l]), owned(blk, dw_ref[..., l]))
        assert e <= tol, (what, l, e)


def check_block_res(engine, dims, prm, seed=1, **mk):
    """blocketteRes core (timeStep + initres + fluxes + sum) vs blockResCore of
    the reference (blockette.F90:755-852)."""
    from oracle import ref
    lvl = new_level(engine)
    prm = prm.replace(currentLevel=lvl, groundLevel=lvl)
    blk = make_block(*dims, prm, seed=seed, **mk)
    r = ref_bind(blk, prm)
    turb = prm.equations == RANSEquations
    ref.block_res_core(True, True, turb)
    engine.set_options(prm)
    engine.register(blk, nn=1, level=lvl)
    engine.blocketteRes(level=lvl, updateIntermed=True, flowRes=True, turbRes=turb)
    dw = engine.download_residual(1, lvl)
    assert_dw(blk, dw, r["dw"], blk.nw)
    for which, name in ((capi.ARR_RADI, "radI"), (capi.ARR_RADJ, "radJ"), (capi.ARR_RADK, "radK"),
                        (capi.ARR_DTL, "dtl")):
        out = np.zeros_like(r[name])
        engine.download_array(which, out, 1, lvl)
        if name == "dtl":   # owned cells carry dt; halos carry the raw inviscid sum
            e = rel_err(out[1:-1, 1:-1, 1:-1], r[name][1:-1, 1:-1, 1:-1])
        else:
            e = rel_err(out, r[name])
        assert e <= TOL, (name, e)
    return blk, r


def check_rk_residual_sequence(engine, dims, prm, seed=3, **mk):
    """residual() inside the RK smoother: rFil = cdisRK(stage+1) with the
    dissipation residual fw PERSISTENT between stages (residuals.F90:61-65,
    fluxes.F90:1085,1193).  The state is perturbed between stages like a real
    stage update would."""
    from oracle import ref
    lvl = new_level(engine)
    prm = prm.replace(currentLevel=lvl, groundLevel=lvl, smoother=RungeKutta)
    blk = make_block(*dims, prm, seed=seed, **mk)
    r = ref_bind(blk, prm)
    engine.set_options(prm)
    engine.register(blk, nn=1, level=lvl)
    ref.call("timeStep_block", 0)
    engine.timeStep(lvl, False)
    rng = np.random.default_rng(seed)
    for stage in range(prm.nRKStages):
        ref.load().ref_set_int(b"rkStage", stage)
        ref.call("initres_flow")
        ref.call("residual_block")
        engine.residual(lvl, stage)
        dw = engine.download_residual(1, lvl)
        assert_dw(blk, dw, r["dw"], 5, what=f"dw stage {stage}")
        fw = np.zeros_like(r["fw"])
        engine.download_array(capi.ARR_FW, fw, 1, lvl)
        assert_dw(blk, fw, r["fw"], 5, what=f"fw stage {stage}")
        # perturb the state identically on both sides
        fac = 1.0 + 1e-3 * rng.uniform(-1, 1, blk["w"].shape[:3])
        for a in (blk, r):
            a["w"][..., 0] *= fac
            a["w"][..., 4] *= fac
            a["p"][...] *= fac
        engine.upload_state(1, lvl)
