"""Parity checks shared by the GPU tests (-m gpu, real MI355X through the C-ABI)
and the CPU-only kernel-logic tests (tests/hostsim emulator).  Every check
compares against the reference's OWN Fortran (oracle/_ref)."""
import itertools

import numpy as np

from adflow_amd import capi
from adflow_amd.params import (FlowParams, EulerEquations, NSEquations, RANSEquations, dissScalar, dissMatrix, upwind,
                               noLimiter, vanAlbeda, minmod, RungeKutta, DADI)
from adflow_amd.synth import make_block
from util import TOL, LOCAL_TOL, rel_err, rel_err_local, owned

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
        el = rel_err_local(owned(blk, dw_gpu[..., l]), owned(blk, dw_ref[..., l]))
        assert el <= max(LOCAL_TOL, 1e4 * tol), (what, "local measure", l, el)


def check_block_res(engine, dims, prm, seed=1, blk=None, **mk):
    """blocketteRes core (timeStep + initres + fluxes + sum) vs blockResCore of
    the reference (blockette.F90:755-852).  blk: a prepared block instead of make_block(dims, ...)."""
    from oracle import ref
    lvl = new_level(engine)
    prm = prm.replace(currentLevel=lvl, groundLevel=lvl)
    if blk is None:
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


def check_foreign_normals_then_own_nodes(engine, dims, prm, seed=5, **mk):
    """Round-4 advisor finding: the face normals replaced from a FOREIGN buffer (sI scaled: no longer metric_block(x)), then the
    nodes uploaded again from the registered array.  The host-side check 'normals equal metric_block(x)' reads the descriptor's
    arrays, which still agree with each other -- but the device holds the foreign sI: kernels that re-form their normals from the
    nodes (SA march, time step) must stay switched off, and the residual must be the reference's with the SCALED sI."""
    from oracle import ref
    lvl = new_level(engine)
    prm = prm.replace(currentLevel=lvl, groundLevel=lvl)
    blk = make_block(*dims, prm, seed=seed, **mk)
    r = ref_bind(blk, prm)
    r["sI"][...] *= 1.03
    turb = prm.equations == RANSEquations
    ref.block_res_core(True, True, turb)
    engine.set_options(prm)
    engine.register(blk, nn=1, level=lvl)
    foreign = np.asfortranarray(blk["sI"] * 1.03)
    engine.upload_array(capi.ARR_SI, foreign, 1, lvl)
    engine.upload_array(capi.ARR_X, blk["x"], 1, lvl)            # the registered pointer: must not re-enable the normals from x
    engine.blocketteRes(level=lvl, updateIntermed=True, flowRes=True, turbRes=turb)
    dw = engine.download_residual(1, lvl)
    assert_dw(blk, dw, r["dw"], blk.nw, what="dw with foreign sI")
    # and back: the own sI uploaded again -> all four device arrays are the descriptor's, the re-formed normals are allowed again
    engine.upload_array(capi.ARR_SI, blk["sI"], 1, lvl)
    r["sI"][...] = blk["sI"]
    ref.block_res_core(True, True, turb)
    engine.blocketteRes(level=lvl, updateIntermed=True, flowRes=True, turbRes=turb)
    assert_dw(blk, engine.download_residual(1, lvl), r["dw"], blk.nw, what="dw with the own sI again")


def check_block_res_vs_blockette(engine, dims, prm, update_intermed=False, seed=1, **mk):
    """adflow_gpu_block_res vs blockette::blocketteResCore (blockette.F90:299-753), the reference's DEFAULT residual path
    (useBlockettes = True, pyADflow.py:5734): metrics recomputed from x per 8^3 tile, fused SA routines, its own timeStep.
    Without updateIntermed only dw is an output of that path; with it also dtl (owned cells) and the spectral radii."""
    from oracle import ref
    lvl = new_level(engine)
    prm = prm.replace(currentLevel=lvl, groundLevel=lvl)
    blk = make_block(*dims, prm, seed=seed, **mk)
    r = ref_bind(blk, prm)
    turb = prm.equations == RANSEquations
    ref.blockette_res_core(update_intermed, True, turb)
    engine.set_options(prm)
    engine.register(blk, nn=1, level=lvl)
    engine.blocketteRes(level=lvl, updateIntermed=update_intermed, flowRes=True, turbRes=turb)
    dw = engine.download_residual(1, lvl)
    assert_dw(blk, dw, r["dw"], blk.nw, what="dw vs blocketteResCore")
    if update_intermed:
        for which, name in ((capi.ARR_RADI, "radI"), (capi.ARR_RADJ, "radJ"), (capi.ARR_RADK, "radK"), (capi.ARR_DTL, "dtl")):
            out = np.zeros_like(r[name])
            engine.download_array(which, out, 1, lvl)
            e = rel_err(out[1:-1, 1:-1, 1:-1], r[name][1:-1, 1:-1, 1:-1])      # copied out for the owned cells / 1..ie (blockette.F90:660-690)
            assert e <= TOL, (name, e)
    return blk, r


def check_brick_block_res(engine, topo, prm, seed=17, fused_halo=False, **mk):
    """The bench's step on a multi-block brick: whalo2 (2-layer exchange, state scrambled before it so that the halos are stale)
    followed by the blocketteRes core with the default flags on EVERY block, against the reference's own whalo2 +
    blocketteResCore per block (blockette.F90:246, 299-753).  fused_halo: ONE call with ADFLOW_RES_HALO (the exchange inside
    blocketteRes, where the library may put the halo-free tiles between the messages' departure and arrival)."""
    from oracle import ref
    blocks, rblocks = setup_brick(engine, topo, prm, seed, **mk)
    rng = np.random.default_rng(seed)
    turb = prm.equations == RANSEquations
    for nn in blocks:
        b, r = blocks[nn], rblocks[nn]
        sl = (slice(2, b.il + 1), slice(2, b.jl + 1), slice(2, b.kl + 1))
        for n in ("w", "p", "rlv", "rev"):
            if n in b.a:
                noise = rng.uniform(0.97, 1.03, b[n][sl].shape)
                b[n][sl] *= noise            # owned cells only: the halos now disagree with their donors
                r[n][...] = b[n]
        engine.upload_state(nn, 1)
    ref.call_level("whalo2", 1, 1, prm.nw)
    if fused_halo:
        engine.blocketteRes(level=1, updateIntermed=False, flowRes=True, turbRes=turb, halo=True)
    else:
        engine.whalo2(1, 1, prm.nw)
        engine.blocketteRes(level=1, updateIntermed=False, flowRes=True, turbRes=turb)
    for nn in sorted(rblocks):
        ref.call_level("setPointers", 1, nn)
        ref.blockette_res_core(False, True, turb)
        dw = engine.download_residual(nn, 1)
        assert_dw(blocks[nn], dw, rblocks[nn]["dw"], blocks[nn].nw, what=f"block {nn}: whalo2 + blocketteResCore")


def setup_brick_with_bc(engine, topo, prm, brick_spec, seed=19, **mk):
    """A brick of blocks whose NON-periodic ends are physical boundaries (brick_spec: {faceID of the brick: BCType}) and whose inner
    faces are 1-to-1 interfaces: boundary subfaces, porosities (setPorosities) and both communication patterns on the engine and in
    the reference's flowDoms.  Returns (blocks, rblocks, bocos)."""
    from oracle import ref
    from adflow_amd.synth import make_bocos, set_porosities
    from adflow_amd.topology import apply_local_copies_fast
    new_level(engine)
    prm = prm.replace(currentLevel=1, groundLevel=1)
    blocks = make_brick(topo, prm, seed, wall_kmin=False, **mk)
    lid = topo.local_ids()
    bocos = {}
    for g in range(topo.nblocks):
        spec = topo.boundary_spec(g, brick_spec)
        nn = lid[g]
        if spec:
            bocos[nn] = make_bocos(blocks[nn], prm, spec, seed=seed + 31 * g + 1)
        set_porosities(blocks[nn], bocos[nn][0] if nn in bocos else [])
    pats = {L: topo.patterns(L)[0] for L in (1, 2)}
    apply_local_copies_fast(blocks, pats[2])
    rblocks = {nn: b.copy() for nn, b in blocks.items()}
    ref.alloc_doms(max(blocks), 1)
    ref.bind_blocks(rblocks, prm, level=1, nlevels=1, alloc=False, bocos=bocos)
    engine.set_options(prm)
    for nn, b in blocks.items():
        engine.register(b, nn=nn, level=1)
        if nn in bocos:
            engine.bc_register(*bocos[nn], nn=nn, level=1)
    for L in (1, 2):
        ref.set_internal_comm(1, L, pats[L])
        engine.comm_register(1, L, pats[L])
    return blocks, rblocks, bocos, prm


def check_blockette_res_with_bc(engine, topo, prm, brick_spec, seed=19, split_eval=None, floor_p=False, allow_degenerate=False, **mk):
    """The reference's WHOLE blocketteRes (blockette.F90:199-283, default flags, storeWall = T) on a wall-bounded mesh, as ONE
    library call (ADFLOW_RES_CLOSURES | HALO | FLOW | TURB): derived values of the owned cells, turbulence + mean-flow boundary
    conditions of every subface, whalo2 over the 1-to-1 interfaces, blocketteResCore; compared: dw of every block and
    viscSubface%tau / %q of every viscous subface.  The state is scrambled first so that p / rlv / rev and every halo are stale."""
    from oracle import ref
    blocks, rblocks, bocos, prm = setup_brick_with_bc(engine, topo, prm, brick_spec, seed, **mk)
    rng = np.random.default_rng(seed)
    turb = prm.equations == RANSEquations
    for nn in sorted(blocks):
        b, r = blocks[nn], rblocks[nn]
        sl = (slice(2, b.il + 1), slice(2, b.jl + 1), slice(2, b.kl + 1))
        b["w"][sl] *= rng.uniform(0.97, 1.03, b["w"][sl].shape)       # owned cells only: closures and halos are now stale
        if floor_p:
            # some owned cells carry less energy than their kinetic energy: computePressureSimple floors their pressure, whalo2 hands
            # the OLD energy to the neighbours' halos and recomputes the owned one from the floored pressure (haloExchange.F90:178-196)
            wo = b["w"][sl]
            sel = rng.uniform(0, 1, wo.shape[:3]) < 0.02
            wo[sel, 4] = 0.4 * wo[sel, 0] * (wo[sel, 1] ** 2 + wo[sel, 2] ** 2 + wo[sel, 3] ** 2)
            b["w"][sl] = wo
        r["w"][...] = b["w"]
        engine.upload_state(nn, 1)
    for nn in sorted(rblocks):
        ref.call_level("setPointers", 1, nn)
        ref.call("computePressureSimple", 0)
        ref.call("computeLamViscosity", 0)
        ref.call("computeEddyViscosity", 0)
        if turb:
            ref.call("bcTurbTreatment")
            ref.call("applyAllTurbBCThisBlock", 1)
        ref.call("applyAllBC_block", 1)
    ref.call_level("whalo2", 1, 1, prm.nw)
    if split_eval is not None:
        engine.set_tuning("split_eval", split_eval)
    try:
        engine.blocketteRes(level=1, updateIntermed=False, flowRes=True, turbRes=turb, halo=True, closures=True)
    finally:
        if split_eval is not None:
            engine.set_tuning("split_eval", 1)
    nwall = 0
    for nn in sorted(rblocks):
        ref.call_level("setPointers", 1, nn)
        ref.blockette_res_core(False, True, turb)
        dw = engine.download_residual(nn, 1)
        if allow_degenerate and not np.isfinite(owned(blocks[nn], rblocks[nn]["dw"])).all():
            # (random sweeps: a floored pressure next to an extrapolation boundary can make the REFERENCE produce NaN; then only the
            # pattern is compared)
            assert np.array_equal(np.isfinite(owned(blocks[nn], dw)), np.isfinite(owned(blocks[nn], rblocks[nn]["dw"])))
            continue
        assert_dw(blocks[nn], dw, rblocks[nn]["dw"], blocks[nn].nw, what=f"block {nn}: blocketteRes with boundary conditions")
        faces, nvisc = bocos.get(nn, ([], 0))
        for mm in range(1, nvisc + 1):
            tau_r, q_r = ref.wall_stress(mm)
            tau, q = engine.wall_stress(tau_r.shape[:2], mm, nn=nn)
            assert np.abs(tau_r).max() > 0
            e = max(rel_err(tau, tau_r), rel_err(q, q_r))
            assert e <= TOL, ("wall stress", nn, mm, faces[mm - 1]["faceID"], e)
            nwall += 1
    return nwall


def check_block_res_approx(engine, dims, prm, diss_approx=True, visc_approx=True, seed=91, blockettes=False, **mk):
    """blockResCore with dissApprox / viscApprox (blockette.F90:755-852): the lumped-dissipation and thin-layer
    residual of the preconditioner assembly, sensor FROZEN at a reference state that differs from the state the
    residual is evaluated at (as in the finite-difference Jacobian, adjointUtils.F90:1909-1969)."""
    from oracle import ref
    lvl = new_level(engine)
    prm = prm.replace(currentLevel=lvl, groundLevel=lvl)
    blk = make_block(*dims, prm, seed=seed, **mk)
    r = ref_bind(blk, prm)
    turb = prm.equations == RANSEquations
    engine.set_options(prm)
    engine.register(blk, nn=1, level=lvl)
    # reference state -> frozen sensor (referenceShockSensor restated: pressure, or entropy for NS/RANS scalar JST)
    if prm.equations == EulerEquations or prm.spaceDiscr == dissMatrix:
        sens = r["p"].copy(order="F")
    else:
        sens = np.asfortranarray(r["p"] / r["w"][..., 0] ** r["gamma"])
    r.a["shockSensor"] = sens
    ref.load().ref_set_ptr(b"shockSensor", sens.ctypes.data)
    engine.referenceShockSensor(lvl)
    # perturbed state (what the coloured finite differences do)
    rng = np.random.default_rng(seed)
    fac = 1.0 + 1e-2 * rng.uniform(-1, 1, blk["w"].shape[:3])
    for a in (blk, r):
        a["w"][..., 0] *= fac
        a["w"][..., 4] *= fac
        a["p"][...] *= fac
    engine.upload_state(1, lvl)
    # the two cores of the reference differ for the Roe scheme: blocketteResCore (the default) is first order here (blockette.F90:643)
    if blockettes:
        ref.blockette_res_core(True, True, turb, diss_approx=diss_approx, visc_approx=visc_approx)
    else:
        ref.block_res_core(True, True, turb, diss_approx=diss_approx, visc_approx=visc_approx)
    engine.blocketteRes(level=lvl, updateIntermed=True, flowRes=True, turbRes=turb, dissApprox=diss_approx, viscApprox=visc_approx,
                        useBlockettes=blockettes)
    dw = engine.download_residual(1, lvl)
    assert_dw(blk, dw, r["dw"], blk.nw, what=f"approx residual diss={diss_approx} visc={visc_approx}")
    # an exact evaluation afterwards must not be affected by the frozen sensor
    ref.block_res_core(True, True, turb)
    engine.blocketteRes(level=lvl, updateIntermed=True, flowRes=True, turbRes=turb)
    dw = engine.download_residual(1, lvl)
    assert_dw(blk, dw, r["dw"], blk.nw, what="exact residual after an approximate one")


def check_fd_jacobian(engine, dims, prm, spec, usePC=True, frozenTurb=False, useTurbOnly=False, viscPC=False, delta=1e-5,
                      tol=1e-9, seed=101, blockettes=False, **mk):
    """adflow_gpu_fd_jacobian vs adjointUtils::setupStateResidualMatrix(useAD=F) (adjointUtils.F90:7-715, restated around the
    reference's own routines in oracle/refbuild/ref_driver.F90:ref_fd_jacobian): every block of the coloured finite-difference
    matrix of ONE block with six physical boundary faces.  A finite difference amplifies the rounding differences between two
    correct residuals by 1/delta, so the tolerance is RELATIVE TO THE LARGEST ENTRY and scales like 1e-15/delta: the 1e-10 bar
    of the residual is met at delta = 1e-5, and the reference's own delta = 1e-9 is checked at the 1e-5 it can resolve."""
    from oracle import ref
    blk, r, prm = setup_block_with_bc(engine, dims, prm, spec, seed, **mk)
    Jr = ref.fd_jacobian(blk.nx, blk.ny, blk.nz, usePC, frozenTurb, useTurbOnly, viscPC, blockettes, delta)
    engine.setupStateResidualMatrix(1, usePC, frozenTurb, useTurbOnly, viscPC, delta)
    ns, st = engine.jacobianInfo()
    Jg = engine.jacobianBlocks(1, 1)
    assert Jg.shape == Jr.shape, (Jg.shape, Jr.shape)
    scale = np.abs(Jr).max()
    assert scale > 0.0
    err = np.abs(Jg - Jr).max() / scale
    assert err <= tol, (err, tol, np.unravel_index(np.abs(Jg - Jr).argmax(), Jr.shape))
    if ns == 6:
        # RANS: next to a wall the SA diagonal dwarfs everything else; the mean-flow blocks against their own scale
        scf = np.abs(Jr[..., :5, :5, :]).max()
        errf = np.abs(Jg[..., :5, :5, :] - Jr[..., :5, :5, :]).max() / scf
        assert errf <= tol, ("mean-flow blocks", errf, tol)
    # the row-ordered download (one contiguous run of blocks per row cell) is the same numbers, permuted
    Jrows = engine.jacobianRows(1, 1)
    assert Jrows.shape == (ns, ns, st.shape[0], blk.nx, blk.ny, blk.nz)
    assert np.array_equal(Jrows, np.transpose(Jg, (3, 4, 5, 0, 1, 2)))
    # no stencil entry silently skipped (the corner entries of the 27-point preconditioner stencil are empty in the reference
    # too: the thin-layer residual does not couple them), the diagonal block of every cell is non-zero
    for s in range(st.shape[0]):
        assert (np.abs(Jg[..., s]).max() > 1e-7 * scale) == (np.abs(Jr[..., s]).max() > 1e-7 * scale), ("stencil entry", s, st[s])
    # (how many entries clear the threshold depends on the dynamic range: next to a wall the SA diagonal is ~1e12 and the
    # off-diagonal blocks fall below 1e-7 of it -- the count has to equal the reference's, not a fixed number)
    n_ref = sum(np.abs(Jr[..., s]).max() > 1e-7 * scale for s in range(st.shape[0]))
    assert sum(np.abs(Jg[..., s]).max() > 1e-7 * scale for s in range(st.shape[0])) == n_ref >= 1
    diag = np.abs(Jg[..., 0 if usePC or prm.equations == EulerEquations else 13]).reshape(blk.nx * blk.ny * blk.nz, -1).max(axis=1)
    assert diag.min() > 0.0
    # resetFDReference: w is the reference state (boundary halos as the reference evaluation left them), dw the scaled residual
    engine.download_state(1, 1)
    assert rel_err(blk["w"], r["w"]) <= TOL
    dw = engine.download_residual(1, 1)
    lo, hi = (5, 6) if useTurbOnly else (0, 5 if frozenTurb else blk.nw)
    assert rel_err(owned(blk, dw)[..., lo:hi], owned(blk, r["dw"])[..., lo:hi]) <= TOL
    return Jg, Jr, st


def check_ad_jacobian(engine, dims, prm, spec, usePC=True, frozenTurb=False, useTurbOnly=False, viscPC=False, tol=1e-10, seed=103, **mk):
    """adflow_gpu_fd_jacobian(ADFLOW_JAC_USE_AD) vs adjointUtils::setupStateResidualMatrix(useAD = T) (adjointUtils.F90:227-409):
    every stencil block of ONE block with six physical boundary faces, the reference side being its own Tapenade forward routines
    (src/adjoint/outputForward) in the call sequence of block_res_state_d (oracle/refbuild/ref_driver.F90:ref_ad_jacobian).  Both
    sides are exact derivatives: the tolerance is the residual's 1e-10, relative to the largest entry -- and, for RANS, of the
    mean-flow blocks against their own scale.  Also: the assembly leaves state and residual of the level untouched."""
    from oracle import ref
    blk, r, prm = setup_block_with_bc(engine, dims, prm, spec, seed, **mk)
    Jr = ref.ad_jacobian(blk.nx, blk.ny, blk.nz, usePC, frozenTurb, useTurbOnly, viscPC)
    w0 = blk["w"].copy(order="F")
    engine.download_state(1, 1)
    w_before = blk["w"].copy(order="F")
    engine.setupStateResidualMatrix(1, usePC, frozenTurb, useTurbOnly, viscPC, useAD=True)
    ns, st = engine.jacobianInfo()
    Jg = engine.jacobianBlocks(1, 1)
    assert Jg.shape == Jr.shape, (Jg.shape, Jr.shape)
    scale = np.abs(Jr).max()
    assert scale > 0.0
    err = np.abs(Jg - Jr).max() / scale
    assert err <= tol, (err, tol, np.unravel_index(np.abs(Jg - Jr).argmax(), Jr.shape))
    if ns == 6:
        scf = np.abs(Jr[..., :5, :5, :]).max()
        errf = np.abs(Jg[..., :5, :5, :] - Jr[..., :5, :5, :]).max() / scf
        assert errf <= tol, ("mean-flow blocks", errf, tol)
    n_ref = sum(np.abs(Jr[..., s]).max() > 1e-7 * scale for s in range(st.shape[0]))
    assert sum(np.abs(Jg[..., s]).max() > 1e-7 * scale for s in range(st.shape[0])) == n_ref >= 1
    engine.download_state(1, 1)
    assert rel_err(blk["w"], w_before) <= 1e-14        # forward mode perturbs nothing (whalo2 in front re-forms the owned energy)
    blk["w"][...] = w0
    return Jg, Jr, st


def check_jacobian_several_blocks(engine, prm, blocks_spec, seed=91, quick=False, **mk):
    """The preconditioner matrix of a LEVEL of several blocks of different sizes (slot numbers with a gap), by finite differences and
    by forward mode: the marching path of round 5 -- k_pc_march + k_sa_march over the level's tile tables, the snapshot entries
    written by the marches through the per-slot table KParams::snapTab at offsets that depend on the block's own box -- against the
    same path with the snapshot kernels and against the kernels it replaced (one launch per block, each of them
    checked against the reference on single blocks in check_fd_jacobian / check_ad_jacobian).  Between two correct assemblies the
    blocks differ by rounding x 1 / delta (finite differences) or by rounding (forward mode)."""
    blocks, rblocks, prm = setup_blocks_with_bc(engine, prm, blocks_spec, seed=seed, **mk)
    keys = {"pc_fused": 1, "jac_snap": 1}

    def assemble(useAD, **tune):
        for k_, v_ in {**keys, **tune}.items():
            engine.set_tuning(k_, v_)
        engine.setupStateResidualMatrix(1, True, delta=1e-6, useAD=useAD)
        return {nn: engine.jacobianBlocks(nn).copy() for nn in blocks}

    try:
        for useAD, tol in ((False, 1e-7), (True, 2e-10)):
            new = assemble(useAD)
            others = {"replaced kernels": assemble(useAD, pc_fused=0, jac_snap=0)}
            if not quick:         # (the emulator twin of the test: the two variants below run on single blocks there)
                others.update({"snapshot kernels": assemble(useAD, jac_snap=0)})
            for nn in blocks:
                scale = np.abs(others["replaced kernels"][nn]).max()
                assert scale > 0.0
                for what, J in others.items():
                    err = np.abs(new[nn] - J[nn]).max() / scale
                    assert err <= tol, (("forward mode" if useAD else "finite differences"), what, nn, err, tol)
    finally:
        for k_, v_ in keys.items():
            engine.set_tuning(k_, v_)


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


# ---------------------------------------------------------------------------
# multi-block checks against the reference's SHELL routines (smoothers.F90,
# haloExchange.F90) on periodic bricks of blocks
# ---------------------------------------------------------------------------
def check_apply_bc(engine, dims, prm, spec, secondHalo=True, seed=51, level=1, split=(), mutate=None, **mk):
    """applyAllBC_block (BCRoutines.F90:57-221) on a block whose six faces are physical boundaries:
    every halo value the reference's routine writes (w, p, gamma, rlv, rev on both halo rings,
    edges and corners included, where later subfaces read what earlier ones wrote)."""
    from oracle import ref
    from adflow_amd.synth import make_bocos
    new_level(engine)
    prm = prm.replace(currentLevel=level, groundLevel=1)
    blk = make_block(*dims, prm, seed=seed, **mk)
    if mutate is not None:
        mutate(blk)
    faces, nvisc = make_bocos(blk, prm, spec, seed=seed + 1, split=split)
    r = blk.copy()
    ref.bind_block(r, prm)
    ref.set_bocos(faces, nvisc)
    ref.call("applyAllBC_block", int(secondHalo))
    engine.set_options(prm)
    engine.register(blk, nn=1, level=level)
    engine.bc_register(faces, nvisc, nn=1, level=level)
    engine.applyAllBC(level, secondHalo)
    assert_state(engine, {1: blk}, {1: r}, prm, f"applyAllBC spec={spec} secondHalo={secondHalo}", level=level)
    g = np.zeros_like(r["gamma"])
    engine.download_array(capi.ARR_GAMMA, g, 1, level)
    assert rel_err(g, r["gamma"]) <= TOL
    return blk, r


def _warp_owned_nodes(blocks, seed):
    """move the nodes 1..il x 1..jl x 1..kl of every block (the halo nodes are what xhalo / exchangeCoor rebuild)"""
    rng = np.random.default_rng(seed)
    for nn in sorted(blocks):
        b = blocks[nn]
        h = 1.0 / max(b.nx, b.ny, b.nz)
        d = 0.05 * h * rng.uniform(-1, 1, b["x"][1:-1, 1:-1, 1:-1].shape)
        b["x"][1:-1, 1:-1, 1:-1] += d
        # poison the halo nodes so that a missing update cannot pass
        for sl in ((0,), (-1,)):
            b["x"][sl[0], :, :] = 7.0; b["x"][:, sl[0], :] = 7.0; b["x"][:, :, sl[0]] = 7.0


def _assert_geometry(engine, blocks, rblocks, what, names=("x", "vol", "sI", "sJ", "sK")):
    ids = {"x": capi.ARR_X, "vol": capi.ARR_VOL, "sI": capi.ARR_SI, "sJ": capi.ARR_SJ, "sK": capi.ARR_SK}
    for nn in sorted(blocks):
        for name in names:
            out = np.zeros_like(rblocks[nn][name])
            engine.download_array(ids[name], out, nn, 1)
            ref_arr = rblocks[nn][name]
            if name == "vol":
                out, ref_arr = out[1:-1, 1:-1, 1:-1], ref_arr[1:-1, 1:-1, 1:-1]
            e = rel_err(out, ref_arr)
            assert e <= TOL, (what, nn, name, e)


def check_coordinate_halos_brick(engine, topo, prm, seed=83, **mk):
    """xhalo_block (adjointExtra.F90:365-599) + exchangeCoor (haloExchange.F90:2456-2640) + volume / metric on a periodic
    brick of blocks after the owned nodes moved: the front part of the `useSpatial` branch of blocketteRes."""
    from oracle import ref
    blocks, rblocks = setup_brick(engine, topo, prm, seed, **mk)
    npat = topo.patterns(0)[0]
    ref.set_internal_comm(1, 0, npat)
    engine.comm_register(1, 0, npat)
    _warp_owned_nodes(blocks, seed)
    for nn in blocks:
        rblocks[nn]["x"][...] = blocks[nn]["x"]
    for nn in sorted(rblocks):
        ref.call_level("setPointers", 1, nn)
        ref.call("xhalo_block")
    ref.call_level("exchangeCoor", 1)
    for nn in sorted(rblocks):
        ref.call_level("setPointers", 1, nn)
        ref.call("volume_block")
        ref.call("metric_block")
    for nn in sorted(blocks):
        engine.upload_coordinates(nn, 1)
    engine.xhalo(1)
    engine.exchangeCoor(1)
    engine.update_geometry(1)
    _assert_geometry(engine, blocks, rblocks, "xhalo + exchangeCoor + metrics")


def periodic_lists(cp, angle=0.3, translation=(0.0, 0.0, 0.4), center=(0.1, -0.2, 0.0)):
    """periodicData of a pattern whose i direction wraps around a rotationally periodic sector: halos that crossed the
    low-i side take the rotation about z by -angle, those that crossed the high-i side by +angle"""
    out = []
    for sign in (-1, 1):
        m = cp.wrapI == sign
        if not m.any():
            continue
        a = sign * angle
        R = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
        out.append(dict(rotMatrix=R, rotCenter=np.array(center), translation=sign * np.array(translation),
                        block=cp.haloBlock[m].copy(), indices=np.asfortranarray(cp.haloIndices[m])))
    return out


def check_periodic_halos(engine, topo, prm, seed=89, **mk):
    """periodic interfaces: correctPeriodicVelocity after whalo1 / whalo2 (haloExchange.F90:456-551) and
    correctPeriodicCoor after exchangeCoor (:2644-2712), applied to the halos listed in periodicData"""
    from oracle import ref
    blocks, rblocks = setup_brick(engine, topo, prm, seed, **mk)
    pats = {L: topo.patterns(L)[0] for L in (0, 1, 2)}
    ref.set_internal_comm(1, 0, pats[0])
    engine.comm_register(1, 0, pats[0])
    for L in (0, 1, 2):
        pl = periodic_lists(pats[L])
        assert pl, "the brick must wrap in i"
        ref.set_periodic(1, L, pl)
        engine.comm_register_periodic(1, L, pl)
    rng = np.random.default_rng(seed)
    for nn in sorted(blocks):
        f = 1.0 + 0.01 * rng.uniform(-1, 1, blocks[nn]["w"].shape)
        for b in (blocks[nn], rblocks[nn]):
            b["w"] *= f
        engine.upload_state(nn, 1)
    for nLayers, name in ((2, "whalo2"), (1, "whalo1")):
        ref.call_level(name, 1, 1, prm.nw)
        getattr(engine, name)(1, 1, prm.nw)
        assert_state(engine, blocks, rblocks, prm, f"{name} with periodic velocity rotation")
    # density only: no rotation (haloExchange.F90:457)
    ref.call_level("whalo2", 1, 1, 1)
    engine.whalo2(1, 1, 1)
    assert_state(engine, blocks, rblocks, prm, "whalo2 of rho only: no periodic correction")
    ref.call_level("exchangeCoor", 1)
    engine.exchangeCoor(1)
    _assert_geometry(engine, blocks, rblocks, "exchangeCoor with periodic transformation", names=("x",))
    for L in (0, 1, 2):        # leave the reference's patterns clean for the next test
        ref.set_periodic(1, L, [])


def check_coarse_level_geometry(engine, topo, prm, seed=87, **mk):
    """updateCoordinatesAllLevels / updateMetricsAllLevels on the device (preprocessingAPI.F90:3945-4017): owned coarse
    nodes by injection (coarseOwnedCoordinates, coarseUtils.F90:780-858, restated with numpy: a pure copy of every
    second node), then xhalo + exchangeCoor + volume / metric of the coarse level against the reference's routines."""
    from oracle import ref
    levels, rlevels = setup_multilevel_brick(engine, topo, prm, nlevels=2, seed=seed, **mk)
    c1 = next(iter(levels[1].values()))
    t2 = topo.coarse() if hasattr(topo, "coarse") else type(topo)(topo.Bi, topo.Bj, topo.Bk, c1.nx, c1.ny, c1.nz)
    for lv, t in ((1, topo), (2, t2)):
        npat = t.patterns(0)[0]
        ref.set_internal_comm(lv, 0, npat)
        engine.comm_register(lv, 0, npat)
    _warp_owned_nodes(levels[0], seed)
    for nn, b in levels[0].items():
        engine.upload_coordinates(nn, 1)
        # numpy statement of coarseOwnedCoordinates: coarse node ii = the fine node that survived the coarsening
        # (2 ii - 1 for an even cell count; the upper fine cell of the coarse cell otherwise, node 1 stays node 1)
        rc = rlevels[1][nn]
        rc["x"][...] = 7.0
        nm = [np.concatenate([[1], rc["mg%sFine" % d][1:-1, 1]]) for d in "IJK"]
        rc["x"][1:-1, 1:-1, 1:-1] = b["x"][np.ix_(nm[0], nm[1], nm[2])]
    p2 = prm.replace(currentLevel=2, groundLevel=1)
    for nn in sorted(rlevels[1]):
        ref.call_level("setPointers", 2, nn)
        ref.call("xhalo_block")
    ref.call_level("exchangeCoor", 2)
    for nn in sorted(rlevels[1]):
        ref.call_level("setPointers", 2, nn)
        ref.call("volume_block")
        ref.call("metric_block")
    engine.xhalo(1)
    engine.exchangeCoor(1)
    engine.coarseOwnedCoordinates(2)
    engine.xhalo(2)
    engine.exchangeCoor(2)
    engine.update_geometry(2)
    ids = {"x": capi.ARR_X, "vol": capi.ARR_VOL, "sI": capi.ARR_SI, "sJ": capi.ARR_SJ, "sK": capi.ARR_SK}
    for nn in sorted(levels[1]):
        for name in ("x", "vol", "sI", "sJ", "sK"):
            out = np.zeros_like(rlevels[1][nn][name])
            engine.download_array(ids[name], out, nn, 2)
            ref_arr = rlevels[1][nn][name]
            if name == "vol":
                out, ref_arr = out[1:-1, 1:-1, 1:-1], ref_arr[1:-1, 1:-1, 1:-1]
            e = rel_err(out, ref_arr)
            assert e <= TOL, ("coarse level geometry", nn, name, e)


def check_xhalo_symmetry(engine, dims, prm, spec, split=(), seed=85, **mk):
    """xhalo_block on a block with symmetry planes (mirror image of the second node plane about BCData%symNorm, node
    ranges extended over the block edges) and other subfaces (plain extrapolation)."""
    from oracle import ref
    from adflow_amd.synth import make_bocos
    new_level(engine)
    prm = prm.replace(currentLevel=1, groundLevel=1)
    blk = make_block(*dims, prm, seed=seed, **mk)
    faces, nvisc = make_bocos(blk, prm, spec, seed=seed + 1, split=split)
    for f in faces:
        if f["bcType"] == -1:      # plane normal: mean of the face normals, deliberately not of unit length
            f["symNorm"] = 1.7 * f["norm"].reshape(-1, 3).mean(axis=0)
    engine.set_options(prm)
    engine.register(blk, nn=1, level=1)
    engine.bc_register(faces, nvisc, nn=1, level=1)
    _warp_owned_nodes({1: blk}, seed)
    r = blk.copy()
    ref.bind_block(r, prm)
    ref.set_bocos(faces, nvisc)
    ref.call("xhalo_block")
    engine.upload_coordinates(1, 1)
    engine.xhalo(1)
    _assert_geometry(engine, {1: blk}, {1: r}, f"xhalo with symmetry planes {spec}", names=("x",))


def check_actuator_regions(engine, dims, prm, seed=93, **mk):
    """sourceTerms_block (residuals.F90:348-425): body force + heat source of actuator regions in `residual`
    (initres; sourceTerms; residual, smoothers.F90:72-75) and after the core of blocketteRes (blockette.F90:276-281),
    with the relaxation ramp between relaxStart and relaxEnd."""
    from oracle import ref
    new_level(engine)
    prm = prm.replace(currentLevel=1, groundLevel=1, ordersConverged=2.5)
    blk = make_block(*dims, prm, seed=seed, **mk)
    rng = np.random.default_rng(seed)
    regions = []
    for m in range(2):
        n = 7 + 5 * m
        ids = np.asfortranarray(np.stack([rng.integers(2, blk.il + 1, n), rng.integers(2, blk.jl + 1, n),
                                          rng.integers(2, blk.kl + 1, n)]).astype(np.int32))
        ids = np.asfortranarray(np.unique(ids, axis=1))          # a cell appears once per region
        regions.append(dict(block=np.ones(ids.shape[1], np.int32), cellIDs=ids, force=prm.pInfDim * rng.uniform(-1, 1, 3),
                            heat=prm.pInfDim * prm.uRef * 0.3, volume=float(blk["vol"][2:-2, 2:-2, 2:-2].sum() * 0.1),
                            relaxStart=(2.0 if m else -1.0), relaxEnd=(3.0 if m else -1.0)))
    r = blk.copy()
    ref.bind_block(r, prm)
    ref.set_actuator_regions(regions)
    engine.set_options(prm)
    engine.register(blk, nn=1, level=1)
    engine.actuator_register(regions)
    try:
        # residual path
        ref.load().ref_set_int(b"rkStage", 0)
        ref.call("timeStep_block", 0)
        ref.call("initres_flow")
        base = r["dw"].copy()
        for m in (1, 2):
            ref.call("sourceTerms_block", m)
        assert np.abs(r["dw"] - base).max() > 0
        ref.call("residual_block")
        engine.timeStep(1, False)
        engine.residual(1, 0)
        assert_dw(blk, engine.download_residual(1, 1), r["dw"], 5, what="residual with actuator sources")
        # blocketteRes path: core, then the sources
        ref.block_res_core(True, True, prm.equations == RANSEquations)
        for m in (1, 2):
            ref.call("sourceTerms_block", m)
        engine.blocketteRes(1, True, True, prm.equations == RANSEquations)
        assert_dw(blk, engine.download_residual(1, 1), r["dw"], blk.nw, what="blocketteRes with actuator sources")
    finally:
        engine.actuator_register([])
        ref.set_actuator_regions([])


def check_wall_stress(engine, dims, prm, spec, split=(), seed=57, dadi=False, **mk):
    """viscSubface(:)%tau / %q: the wall stress tensor and heat flux viscousFlux stores for the viscous subfaces when
    rkStage == 0 on the ground level (fluxes.F90:2586-2592, 2861-2892 k, 3155-3185 j, 3450-3480 i)."""
    from oracle import ref
    from adflow_amd.synth import make_bocos
    new_level(engine)
    prm = prm.replace(currentLevel=1, groundLevel=1)
    blk = make_block(*dims, prm, seed=seed, **mk)
    faces, nvisc = make_bocos(blk, prm, spec, seed=seed + 1, split=split)
    assert nvisc > 0
    r = blk.copy()
    ref.bind_block(r, prm)
    ref.set_bocos(faces, nvisc)
    ref.call("applyAllBC_block", 1)
    ref.load().ref_set_int(b"rkStage", 0)
    ref.call("timeStep_block", 0)
    ref.call("initres_flow")
    ref.call("residual_block")
    engine.set_options(prm)
    engine.register(blk, nn=1, level=1)
    engine.bc_register(faces, nvisc, nn=1, level=1)
    engine.applyAllBC(1, True)
    engine.timeStep(1, False)
    engine.residual(1, 0)
    for mm in range(1, nvisc + 1):
        tau_r, q_r = ref.wall_stress(mm)
        tau, q = engine.wall_stress(tau_r.shape[:2], mm)
        assert np.abs(tau_r).max() > 0
        e = max(rel_err(tau, tau_r), rel_err(q, q_r))
        assert e <= TOL, (mm, faces[mm - 1]["faceID"], e)
    dw = engine.download_residual(1, 1)
    assert_dw(blk, dw, r["dw"], 5, what="dw with wall stress storage")


def check_update_geometry(engine, dims, prm, spec, seed=81, **mk):
    """volume_block + metric_block + boundaryNormals (adjointExtra.F90:5-364) after the nodes moved: vol, sI/sJ/sK
    and - through a boundary-condition pass that reads them - the unit normals of the boundary subfaces."""
    from oracle import ref
    from adflow_amd.synth import make_bocos
    new_level(engine)
    prm = prm.replace(currentLevel=1, groundLevel=1)
    blk = make_block(*dims, prm, seed=seed, **mk)
    faces, nvisc = make_bocos(blk, prm, spec, seed=seed + 1)
    engine.set_options(prm)
    engine.register(blk, nn=1, level=1)
    engine.bc_register(faces, nvisc, nn=1, level=1)
    # warp the mesh (same on both sides), then derive the metrics
    rng = np.random.default_rng(seed)
    h = 1.0 / max(dims)
    blk["x"] += 0.05 * h * rng.uniform(-1, 1, blk["x"].shape)
    r = blk.copy()
    rfaces = [dict(f, norm=f["norm"].copy(order="F")) for f in faces]
    ref.bind_block(r, prm)
    ref.set_bocos(rfaces, nvisc)
    ref.call("volume_block")
    ref.call("metric_block")
    ref.call("boundaryNormals")
    engine.upload_coordinates(1, 1)
    engine.update_geometry(1)
    for which, name in ((capi.ARR_VOL, "vol"), (capi.ARR_SI, "sI"), (capi.ARR_SJ, "sJ"), (capi.ARR_SK, "sK")):
        out = np.zeros_like(r[name])
        engine.download_array(which, out, 1, 1)
        ref_arr = r[name]
        if name == "vol":      # the reference's loops leave the outermost ring at the `vol = zero` it starts from
            assert np.all(out[0] == 0.0) and np.all(out[:, 0] == 0.0) and np.all(out[:, :, 0] == 0.0)
            out, ref_arr = out[1:-1, 1:-1, 1:-1], ref_arr[1:-1, 1:-1, 1:-1]
        e = rel_err(out, ref_arr)
        assert e <= TOL, (name, e)
    ref.call("applyAllBC_block", 1)
    engine.applyAllBC(1, True)
    assert_state(engine, {1: blk}, {1: r}, prm, "BCs with the recomputed boundary normals")


def check_wall_distance(engine, dims, prm, seed=83, **mk):
    """wallDistance::updateWallDistancesQuickly (wallDistance.F90:36-120) after a mesh warp: d2Wall of the owned cells from the
    wall association (four surface nodes + (u, v) per cell, cells without a wall in reach = large) and the moved surface nodes."""
    from oracle import ref
    new_level(engine)
    prm = prm.replace(currentLevel=1, groundLevel=1)
    blk = make_block(*dims, prm, seed=seed, **mk)
    engine.set_options(prm)
    engine.register(blk, nn=1, level=1)
    rng = np.random.default_rng(seed)
    nsurf = 57
    ind = np.asfortranarray(rng.integers(1, nsurf + 1, size=(4, blk.nx, blk.ny, blk.nz), dtype=np.int32))
    ind[0][rng.uniform(size=ind.shape[1:]) < 0.1] = 0                       # too far away: no association
    uv = np.asfortranarray(rng.uniform(0.0, 1.0, size=(2, blk.nx, blk.ny, blk.nz)))
    uv[:, 0, 0, 0] = (0.0, 1.0)
    engine.registerWallAssociation(ind, uv)
    # warp the mesh and the surface, then update
    h = 1.0 / max(dims)
    blk["x"] += 0.05 * h * rng.uniform(-1, 1, blk["x"].shape)
    xSurf = rng.uniform(-0.2, 1.2, size=3 * nsurf)
    r = blk.copy()
    ref.alloc_doms(1, 1)
    ref.bind_block(r, prm)
    r["d2Wall"][...] = -1.0
    ref.update_wall_distances(ind, uv, xSurf)
    engine.upload_coordinates(1, 1)
    engine.updateWallDistancesQuickly(xSurf, 1)
    out = np.zeros_like(r["d2Wall"])
    engine.download_array(capi.ARR_D2WALL, out, 1, 1)
    assert (r["d2Wall"] >= 1e37).sum() == (ind[0] == 0).sum() > 0
    assert np.array_equal(out >= 1e37, r["d2Wall"] >= 1e37)
    far = r["d2Wall"] >= 1e37
    assert rel_err(np.where(far, 0.0, out), np.where(far, 0.0, r["d2Wall"])) <= TOL
    assert r["d2Wall"].min() > 0.0


def check_smoother_with_bc(engine, dims, prm, spec, seed=61, nsweeps=2, **mk):
    """RungeKuttaSmoother / DADISmoother on ONE block whose six faces are physical boundaries: the device
    applies applyAllBC between update and halo exchange (smoothers.F90:369,680) exactly where the reference does."""
    from oracle import ref
    blk, r, prm = setup_block_with_bc(engine, dims, prm, spec, seed, **mk)
    name = "RungeKuttaSmoother" if prm.smoother == RungeKutta else "DADISmoother"
    for sweep in range(nsweeps):
        ref.load().ref_set_int(b"rkStage", 0)
        ref.call_level("timeStep", 1, 0)
        ref.call_level("initres", 1, 1, 5)
        ref.call_level("residual", 1)
        ref.call_level(name, 1)
        engine.timeStep(1, False)
        engine.residual(1, 0)
        getattr(engine, name)(1)
        assert_state(engine, {1: blk}, {1: r}, prm, f"{name} with BCs, sweep {sweep}")


def make_brick(topo, prm, seed=1, rank=0, **mk):
    """Blocks of `rank` in a BrickTopology with halos made consistent by the
    same-process copy lists (valid when all blocks live on one rank)."""
    from adflow_amd.topology import apply_local_copies_fast
    lid = topo.local_ids()
    blocks = {}
    for g in topo.blocks_of(rank):
        if hasattr(topo, "make_block"):       # LatticeTopology: blocks of different sizes / orientations inside one analytic map
            blocks[lid[g]] = topo.make_block(g, prm, seed=seed + 17 * g, **mk)
        else:
            blocks[lid[g]] = make_block(topo.nx, topo.ny, topo.nz, prm, seed=seed + 17 * g, **mk)
    return blocks


def setup_brick(engine, topo, prm, seed=1, **mk):
    """Register a single-rank brick on the engine and in the reference's
    flowDoms; returns (gpu blocks, reference blocks)."""
    from oracle import ref
    from adflow_amd.topology import apply_local_copies_fast
    lvl = new_level(engine)
    blocks = make_brick(topo, prm, seed, **mk)
    pats = {L: topo.patterns(L)[0] for L in (1, 2)}
    apply_local_copies_fast(blocks, pats[2])
    rblocks = {nn: b.copy() for nn, b in blocks.items()}
    ref.bind_blocks(rblocks, prm.replace(currentLevel=1, groundLevel=1))
    for L in (1, 2):
        ref.set_internal_comm(1, L, pats[L])
    engine.set_options(prm)
    for nn, b in blocks.items():
        engine.register(b, nn=nn, level=lvl)
    for L in (1, 2):
        engine.comm_register(lvl, L, pats[L])
    return blocks, rblocks


def assert_state(engine, blocks, rblocks, prm, what, tol=TOL, level=1, rlv_first_halo_only=False, rlv_no_edges=False, nonfinite_ok=False):
    """rlv_no_edges: leave the halo EDGES and corners of rlv out (cells outside the owned range in two or more directions): with
    several blocks AND boundary subfaces the reference's coarse levels -- whose rlv pointer aliases the FINE array (utils.F90:3420) --
    exchange / extrapolate into such cells of the fine array; no stencil of an owned cell reads rlv there (the viscous flux averages
    it across a face)."""
    names = ["w", "p"] + (["rlv"] if prm.viscous else []) + (["rev"] if prm.eddyModel else [])
    for nn, b in blocks.items():
        engine.download_state(nn, level)
        for n in names:
            a, r = b[n], rblocks[nn][n]
            if n == "rlv" and rlv_no_edges:
                out = [(np.arange(m) < 2) | (np.arange(m) > m - 3) for m in a.shape]
                edge = (out[0][:, None, None].astype(int) + out[1][None, :, None] + out[2][None, None, :]) >= 2
                a, r = np.where(edge, 0.0, a), np.where(edge, 0.0, r)
            if n == "rlv" and rlv_first_halo_only:
                a, r = a[1:-1, 1:-1, 1:-1], r[1:-1, 1:-1, 1:-1]
            if nonfinite_ok and not (np.isfinite(a).all() and np.isfinite(r).all()):
                # (random sweeps: inf / NaN that the REFERENCE produces too, e.g. in halo corners of degenerate boundary sets: the same
                # cells must be non-finite on both sides, the others are compared)
                assert np.array_equal(np.isfinite(a), np.isfinite(r)), (what, nn, n, "non-finite pattern")
                fin = np.isfinite(r)
                a, r = np.where(fin, a, 0.0), np.where(fin, r, 0.0)
            if n == "w":
                for l in range(b.nw):
                    e = rel_err(a[..., l], r[..., l])
                    assert e <= tol, (what, nn, n, l, e)
            else:
                e = rel_err(a, r)
                assert e <= tol, (what, nn, n, e)


def check_halo_exchange(engine, topo, prm, nLayers=2, seed=5):
    """whalo1 / whalo2 over same-process block pairs (haloExchange.F90:657-678)."""
    from oracle import ref
    blocks, rblocks = setup_brick(engine, topo, prm, seed)
    rng = np.random.default_rng(seed)
    # scramble the owned cells identically, poison the halos differently, then exchange
    for nn in blocks:
        b, r = blocks[nn], rblocks[nn]
        for n in ("w", "p", "rlv", "rev"):
            noise = rng.uniform(0.9, 1.1, b[n].shape)
            b[n][...] *= noise
            r[n][...] = b[n]
        engine.upload_state(nn, 1)
    nwf = 5
    ref.call_level("whalo2" if nLayers == 2 else "whalo1", 1, 1, nwf)
    (engine.whalo2 if nLayers == 2 else engine.whalo1)(1, 1, nwf)
    # copies are exact; whalo2 also recomputes rhoE of the owned cells (arithmetic, FMA-contraction level)
    assert_state(engine, blocks, rblocks, prm, f"whalo{nLayers}", tol=1e-14)


def check_halo_loopback(engine, topo, nranks, prm, nLayers=2, seed=5):
    """The inter-rank leg of whalo1 / whalo2 (haloExchange.F90:553-719: pack -> isend / irecv -> unpack) on ONE device: the
    brick is split over `nranks` virtual ranks; for every rank in turn the library gets THAT rank's commPatternCell / internalCell
    lists (block ids mapped onto the ids of the blocks resident here), runs its same-process copies and packs every send slot
    (k_halo_pack); then every rank unpacks the messages addressed to it (k_halo_unpack).  The result must equal the reference's
    own whalo on the undivided brick.  On the MI355X this runs the HIP pack / unpack kernels the RCCL path launches."""
    import copy
    import ctypes
    from oracle import ref
    from adflow_amd.topology import BrickTopology
    blocks, rblocks = setup_brick(engine, topo, prm, seed)
    rng = np.random.default_rng(seed)
    for nn in blocks:
        b, r = blocks[nn], rblocks[nn]
        for n in ("w", "p", "rlv", "rev"):
            noise = rng.uniform(0.9, 1.1, b[n].shape)
            b[n][...] *= noise
        # whalo2 closes with computeEtotBlock on the owned cells (haloExchange.F90:177-197); this check only runs the
        # transport, so the total energy is made consistent with p beforehand (the closing step is then a no-op)
        w = b["w"]
        w[..., 4] = b["p"] / (prm.gammaConstant - 1.0) + 0.5 * w[..., 0] * (w[..., 1] ** 2 + w[..., 2] ** 2 + w[..., 3] ** 2)
        for n in ("w", "p", "rlv", "rev"):
            r[n][...] = b[n]
        engine.upload_state(nn, 1)
    nwf = 5
    ref.call_level("whalo2" if nLayers == 2 else "whalo1", 1, 1, nwf)
    split = (topo.with_owner(lambda g: g % nranks) if hasattr(topo, "with_owner")
             else BrickTopology(topo.Bi, topo.Bj, topo.Bk, topo.nx, topo.ny, topo.nz, owner=lambda g: g % nranks))
    lid1, lidr = topo.local_ids(), split.local_ids()
    pats = split.patterns(nLayers)
    var = (1, nwf, 1, 1)
    nvar = nwf + 1 + 2
    lib = engine.lib

    def resident(cp, rank):
        """the rank's pattern with its local block ids replaced by the ids of the same blocks on this device"""
        m = np.zeros(split.nblocks + 2, np.int32)
        for g in split.blocks_of(rank):
            m[lidr[g]] = lid1[g]
        c = copy.copy(cp)
        for n in ("donorBlock", "haloBlock", "sendBlock", "recvBlock"):
            setattr(c, n, m[getattr(cp, n)].astype(np.int32))
        return c

    msgs = {}
    for r in range(nranks):
        cp = resident(pats[r], r)
        engine.comm_register(1, nLayers, cp)
        capi.check(lib.adflow_gpu_halo_local_copy(1, nLayers, *var), lib)
        for s in range(len(cp.sendProc)):
            peer, cnt = ctypes.c_int(), ctypes.c_int()
            capi.check(lib.adflow_gpu_halo_slot_info(1, nLayers, 1, s, ctypes.byref(peer), ctypes.byref(cnt)), lib)
            buf = np.zeros(nvar * cnt.value)
            capi.check(lib.adflow_gpu_halo_pack(1, nLayers, s, *var, buf.ctypes.data), lib)
            msgs[(r, peer.value)] = buf
    assert msgs, "the split must produce inter-rank messages"
    for r in range(nranks):
        cp = resident(pats[r], r)
        engine.comm_register(1, nLayers, cp)
        for q in range(len(cp.recvProc)):
            peer, cnt = ctypes.c_int(), ctypes.c_int()
            capi.check(lib.adflow_gpu_halo_slot_info(1, nLayers, 0, q, ctypes.byref(peer), ctypes.byref(cnt)), lib)
            buf = msgs[(peer.value, r)]
            assert buf.size == nvar * cnt.value
            capi.check(lib.adflow_gpu_halo_unpack(1, nLayers, q, *var, buf.ctypes.data), lib)
    assert_state(engine, blocks, rblocks, prm, f"whalo{nLayers} through pack / unpack of {nranks} virtual ranks", tol=1e-14)


def check_vacuum_smoother(engine, topo, prm, seed=7, frac=2e-4, **mk):
    """RungeKuttaSmoother on a brick whose blocks hold a near-vacuum pocket: the stage update must clip density and pressure
    at 1e-4 of the free stream exactly as executeRkStage does (smoothers.F90:326, 342).  Returns the number of cells the
    REFERENCE clipped (counted on its arrays after the sweep)."""
    from oracle import ref
    from adflow_amd.topology import apply_local_copies_fast
    import adversarial
    prm = prm.replace(smoother=RungeKutta)
    lvl = new_level(engine)
    blocks = make_brick(topo, prm, seed, **mk)
    for b in blocks.values():
        adversarial.vacuum_pocket(b, prm, frac)
    pats = {L: topo.patterns(L)[0] for L in (1, 2)}
    apply_local_copies_fast(blocks, pats[2])
    rblocks = {nn: b.copy() for nn, b in blocks.items()}
    ref.bind_blocks(rblocks, prm.replace(currentLevel=1, groundLevel=1))
    engine.set_options(prm)
    for L in (1, 2):
        ref.set_internal_comm(1, L, pats[L])
    for nn, b in blocks.items():
        engine.register(b, nn=nn, level=lvl)
    for L in (1, 2):
        engine.comm_register(lvl, L, pats[L])
    ref.load().ref_set_int(b"rkStage", 0)
    ref.call_level("timeStep", 1, 0)
    ref.call_level("initres", 1, 1, 5)
    ref.call_level("residual", 1)
    ref.call_level("RungeKuttaSmoother", 1)
    engine.timeStep(1, False)
    engine.residual(1, 0)
    engine.RungeKuttaSmoother(1)
    clipped = 0
    for r in rblocks.values():
        s = (slice(2, r.il + 1), slice(2, r.jl + 1), slice(2, r.kl + 1))
        clipped += int((r["w"][s + (0,)] == 1.e-4 * prm.rhoInf).sum()) + int((r["p"][s] == 1.e-4 * prm.pInfCorr).sum())
    assert_state(engine, blocks, rblocks, prm, "RK sweep over a near-vacuum pocket")
    return clipped


def check_rk_smoother(engine, topo, prm, seed=7, nsweeps=1, **mk):
    """RungeKuttaSmoother (smoothers.F90:4-88) incl. halo exchange between stages,
    on a periodic brick (no physical boundaries)."""
    from oracle import ref
    prm = prm.replace(smoother=RungeKutta)
    blocks, rblocks = setup_brick(engine, topo, prm, seed, **mk)
    for sweep in range(nsweeps):
        ref.load().ref_set_int(b"rkStage", 0)
        ref.call_level("timeStep", 1, 0)
        ref.call_level("initres", 1, 1, 5)
        ref.call_level("residual", 1)
        ref.call_level("RungeKuttaSmoother", 1)
        engine.timeStep(1, False)
        engine.residual(1, 0)
        engine.RungeKuttaSmoother(1)
        assert_state(engine, blocks, rblocks, prm, f"RK sweep {sweep}")


def check_dadi_smoother(engine, topo, prm, seed=9, nsweeps=1, **mk):
    """DADISmoother (smoothers.F90:383-693, computedwDADI residuals.F90:1062)."""
    from oracle import ref
    prm = prm.replace(smoother=DADI)
    blocks, rblocks = setup_brick(engine, topo, prm, seed, **mk)
    for sweep in range(nsweeps):
        ref.load().ref_set_int(b"rkStage", 0)
        ref.call_level("timeStep", 1, 0)
        ref.call_level("initres", 1, 1, 5)
        ref.call_level("residual", 1)
        ref.call_level("DADISmoother", 1)
        engine.timeStep(1, False)
        engine.residual(1, 0)
        engine.DADISmoother(1)
        assert_state(engine, blocks, rblocks, prm, f"DADI sweep {sweep}")


def setup_multilevel_brick(engine, topo, prm, nlevels=2, seed=1, bc_spec=None, bc_split=None, brick_spec=None, **mk):
    """Periodic bricks on levels 1..nlevels (2:1 coarsening) on the engine and in the
    reference's flowDoms, with 1-to-1 patterns on every level.
    bc_spec (single block only): the six faces are physical boundaries on every level instead.
    bc_split ({faceID: BCType of the upper half}): those faces carry TWO subfaces; the node where they meet survives every
    coarsening (createCoarseBlocks keeps subface boundaries, coarseUtils.F90:117-127), which makes coarse cells of ONE fine
    cell with restriction weight 1/2 in the interior of the block.
    brick_spec (any number of blocks): the faces on the outside of a non-periodic topology are physical boundaries of the kinds
    topo.boundary_spec gives, the faces inside stay 1-to-1 interfaces -- on every level.
    Blocks with odd cell counts coarsen irregularly (coarseUtils.F90:134-153, 281-295).
    Returns (levels, rlevels): lists of {nn: Block}, index 0 = level 1."""
    from oracle import ref
    from adflow_amd.synth import make_coarse_block, make_bocos
    from adflow_amd.topology import BrickTopology, CommPattern, apply_local_copies_fast
    engine.release_all()
    levels = [make_brick(topo, prm, seed, **mk)]
    topos = [topo]
    # cut of a split face: last cell of the lower subface along the face's first index = the NODE that must survive
    cuts = [{}]                                             # per level {faceID: h}
    for fid in dict(bc_split or {}):
        n1 = topo.ny if fid <= 2 else topo.nx               # first index of the face: j on i faces, i on j and k faces
        cuts[0][fid] = (1 + n1 + 2) // 2
    for lv in range(2, nlevels + 1):
        t = topos[-1]
        keep = [[], [], []]
        for fid, h in cuts[-1].items():
            keep[1 if fid <= 2 else 0].append(h)
        levels.append({nn: make_coarse_block(b, prm, keep=keep, seed=seed + 1000 * lv + nn, **mk) for nn, b in levels[-1].items()})
        nm = next(iter(levels[-1].values())).nodeMap
        cuts.append({fid: int(np.where(nm[1 if fid <= 2 else 0] == h)[0][0]) + 1 for fid, h in cuts[-1].items()})
        c1 = next(iter(levels[-1].values()))       # equal blocks coarsen alike (odd counts: createCoarseBlocks' irregular coarsening)
        topos.append(t.coarse() if hasattr(t, "coarse") else BrickTopology(t.Bi, t.Bj, t.Bk, c1.nx, c1.ny, c1.nz))
    bocos = [None] * nlevels
    if bc_spec:
        assert topo.nblocks == 1
        pats = [{L: CommPattern() for L in (1, 2)} for t in topos]
        bocos = [{1: make_bocos(lev[1], prm, bc_spec, seed=seed + 7 * lv, split=bc_split or (), split_at=cuts[lv])}
                 for lv, lev in enumerate(levels)]
    else:
        pats = [{L: t.patterns(L)[0] for L in (1, 2)} for t in topos]
        apply_local_copies_fast(levels[0], pats[0][2])
        if brick_spec:
            from adflow_amd.synth import set_porosities
            bocos = []
            for lv, (t, lev) in enumerate(zip(topos, levels)):
                lid = t.local_ids()
                bl = {}
                for g in range(t.nblocks):
                    spec = t.boundary_spec(g, brick_spec)
                    if spec:
                        bl[lid[g]] = make_bocos(lev[lid[g]], prm, spec, seed=seed + 7 * lv + 31 * g)
                    set_porosities(lev[lid[g]], bl[lid[g]][0] if lid[g] in bl else [])
                bocos.append(bl)
    rlevels = [{nn: b.copy() for nn, b in lev.items()} for lev in levels]
    p1 = prm.replace(currentLevel=1, groundLevel=1)
    for lv, rl in enumerate(rlevels, start=1):
        ref.bind_blocks(rl, p1, level=lv, nlevels=nlevels, alloc=(lv == 1), bocos=bocos[lv - 1])
    engine.set_options(prm)
    for lv, lev in enumerate(levels, start=1):
        for nn, b in lev.items():
            engine.register(b, nn=nn, level=lv)
            if bocos[lv - 1] and nn in bocos[lv - 1]:
                engine.bc_register(*bocos[lv - 1][nn], nn=nn, level=lv)
    for lv in range(1, nlevels + 1):
        for L in (1, 2):
            ref.set_internal_comm(lv, L, pats[lv - 1][L])
            engine.comm_register(lv, L, pats[lv - 1][L])
    if bc_spec or brick_spec:
        ref.call_level("applyAllBC", 1, 1)
        engine.applyAllBC(1, True)
    return levels, rlevels


def count_half_weight_cells(levels):
    """(coarse cells with restriction weight 1/2 at a block end, in the interior of a direction) summed over the coarse levels:
    lets a test assert that it really runs createCoarseBlocks' irregular coarsening"""
    ends = inner = 0
    for lev in levels[1:]:
        b = next(iter(lev.values()))
        for d in "IJK":
            w = b["mg%sWeight" % d]
            half = np.where(w == 0.5)[0]
            ends += int(((half == 0) | (half == w.size - 1)).sum())
            inner += int(((half > 0) & (half < w.size - 1)).sum())
    return ends, inner


def setup_two_level_brick(engine, topo, prm, seed=1, **mk):
    levels, rlevels = setup_multilevel_brick(engine, topo, prm, 2, seed, **mk)
    return levels[0], levels[1], rlevels[0], rlevels[1]


def check_mg_transfer(engine, topo, prm, seed=11, irregular=None, **mk):
    """transferToCoarseGrid then transferToFineGrid (multiGrid.F90:5-652).
    irregular: (half-weight cells at block ends, in the interior) the coarsening must produce (count_half_weight_cells)."""
    from oracle import ref
    fine, coarse, rfine, rcoarse = setup_two_level_brick(engine, topo, prm, seed, **mk)
    if irregular is not None:
        assert count_half_weight_cells([fine, coarse]) == tuple(irregular)
    ref.load().ref_set_int(b"rkStage", 0)
    ref.call_level("transferToCoarseGrid", 1)
    engine.transferToCoarseGrid(1)
    assert_state(engine, coarse, rcoarse, prm, "restricted state", level=2)
    for nn, c in coarse.items():
        for which, name in ((capi.ARR_WR, "wr"), (capi.ARR_W1, "w1"), (capi.ARR_P1, "p1")):
            out = np.zeros_like(rcoarse[nn][name])
            engine.download_array(which, out, nn, 2)
            e = rel_err(out, rcoarse[nn][name])
            assert e <= TOL, ("coarse", nn, name, e)
    # a coarse-level smoothing sweep changes the coarse state, then prolongate the corrections
    ref.call_level("RungeKuttaSmoother" if prm.smoother == RungeKutta else "DADISmoother", 2)
    (engine.RungeKuttaSmoother if prm.smoother == RungeKutta else engine.DADISmoother)(2)
    assert_state(engine, coarse, rcoarse, prm, "coarse smoothing", level=2)
    ref.call_level("transferToFineGrid", 1, 1)
    engine.transferToFineGrid(1)
    assert_state(engine, fine, rfine, prm, "prolongated state", level=1)


def check_mg_cycle(engine, topo, prm, cycling, ncycles=2, seed=13, nlevels=2, bc_spec=None, bc_split=None, irregular=None,
                   brick_spec=None, allow_degenerate=False, **mk):
    """executeMGCycle (multiGrid.F90:825-955) for a given cycling strategy.
    irregular: (half-weight cells at block ends, in the interior) the coarsening must produce (count_half_weight_cells)."""
    from oracle import ref
    levels, rlevels = setup_multilevel_brick(engine, topo, prm, nlevels, seed, bc_spec=bc_spec, bc_split=bc_split, brick_spec=brick_spec,
                                             **mk)
    if irregular is not None:
        assert count_half_weight_cells(levels) == tuple(irregular), count_half_weight_cells(levels)
    fine, rfine = levels[0], rlevels[0]
    ref.set_cycling(cycling)
    # entry condition of the cycle: time step and residual of the ground level are known
    ref.load().ref_set_int(b"rkStage", 0)
    ref.call_level("timeStep", 1, 0)
    ref.call_level("initres", 1, 1, 5)
    ref.call_level("residual", 1)
    engine.timeStep(1, False)
    engine.residual(1, 0)
    for n in range(ncycles):
        ref.call_level("executeMGCycle", 1)
        engine.executeMGCycle(cycling)
        # With physical boundaries the reference's coarse levels scribble over the corner of the FINE rlv array
        # (setPointers aliases rlv to level 1, utils.F90:3420) and the symmetry 2nd-halo pass then copies such a
        # value into 2nd-halo EDGE cells before the wall/farfield pass repairs its source.  No stencil of an owned
        # cell reads those cells; every level owns its rlv here, so they are left out of the comparison.
        assert_state(engine, fine, rfine, prm, f"MG cycle {n}", level=1, rlv_first_halo_only=bool(bc_spec or brick_spec),
                     rlv_no_edges=bool(brick_spec), nonfinite_ok=allow_degenerate)
        for nn, b in fine.items():
            dw = engine.download_residual(nn, 1)
            if allow_degenerate and not np.isfinite(owned(b, rfine[nn]["dw"])).all():
                # (random sweeps: a cycle on a tiny coarse level can make the REFERENCE produce NaN; then only the pattern is compared)
                assert np.array_equal(np.isfinite(owned(b, dw)), np.isfinite(owned(b, rfine[nn]["dw"])))
                return
            assert_dw(b, dw, rfine[nn]["dw"], 5, what=f"residual after cycle {n}")


def check_nk_residual(engine, topo, prm, seed=21, bc_spec=None, floor_p=False, **mk):
    """FormFunction_mf = setW + blocketteRes + setRVec (NKSolvers.F90:437-461,1262-1376):
    the vector glue is restated in numpy (NKSolvers.F90 needs PETSc), every
    arithmetic step in between is the reference's own routine.
    floor_p: some cells of the vector carry so little energy that computePressureSimple floors their pressure: whalo2 then hands the
    VECTOR's energy to the neighbours' halos and recomputes the owned one from the floored pressure (haloExchange.F90:178-196)"""
    from oracle import ref
    if bc_spec:      # one block, physical boundaries applied on the device inside blocketteRes
        blk, r, prm = setup_block_with_bc(engine, (topo.nx, topo.ny, topo.nz), prm, bc_spec, seed, **mk)
        blocks, rblocks = {1: blk}, {1: r}
    else:
        blocks, rblocks = setup_brick(engine, topo, prm, seed, **mk)
    rng = np.random.default_rng(seed)
    nw = prm.nw
    # state vector in PETSc order: block, k, j, i, variable fastest
    parts = []
    for nn in sorted(blocks):
        b = blocks[nn]
        wv = np.ascontiguousarray(np.transpose(b.owned("w"), (2, 1, 0, 3))).reshape(-1, nw).copy()
        wv *= 1.0 + 1e-3 * rng.uniform(-1, 1, wv.shape)
        if nw > 5:
            wv[::7, 5] = 0.0       # exercises the 1e-6*wInf clipping of setW
        if floor_p:
            sel = slice(3, None, 5)
            wv[sel, 4] = 0.4 * wv[sel, 0] * (wv[sel, 1] ** 2 + wv[sel, 2] ** 2 + wv[sel, 3] ** 2)     # below the kinetic energy: p < 0
        parts.append(wv.reshape(-1))
    wVec = np.concatenate(parts)
    # --- reference side
    winf = prm.wInf()
    off = 0
    for nn in sorted(rblocks):
        r = rblocks[nn]
        n = r.ncells * nw
        wv = wVec[off:off + n].reshape(r.nz, r.ny, r.nx, nw)
        off += n
        r.owned("w")[...] = np.transpose(wv, (2, 1, 0, 3))
        if nw > 5:
            r.owned("w")[..., 5] = np.maximum(1e-6 * winf[5], r.owned("w")[..., 5])
        ref.call_level("setPointers", 1, nn)
        ref.call("computePressureSimple", 0)
        ref.call("computeLamViscosity", 0)
        ref.call("computeEddyViscosity", 0)
        if bc_spec:     # blockette.F90:218-226
            if nw > 5:
                ref.call("bcTurbTreatment")
                ref.call("applyAllTurbBCThisBlock", 1)
            ref.call("applyAllBC_block", 1)
    ref.call_level("whalo2", 1, 1, nw)
    rparts = []
    for nn in sorted(rblocks):
        r = rblocks[nn]
        ref.call_level("setPointers", 1, nn)
        ref.block_res_core(False, True, prm.eddyModel)
        res = r.owned("dw") / r.owned("volRef")[..., None]
        if nw > 5:
            res[..., 5] *= prm.turbResScale
        rparts.append(np.ascontiguousarray(np.transpose(res, (2, 1, 0, 3))).reshape(-1))
    rRef = np.concatenate(rparts)
    # --- GPU side
    rGpu = engine.FormFunction_mf(wVec)
    rr = rGpu.reshape(-1, nw)
    rf = rRef.reshape(-1, nw)
    for l in range(nw):
        e = rel_err(rr[:, l], rf[:, l])
        assert e <= TOL, ("rVec", l, e)
    # getRes (no turbResScale) and setRVec norms
    r2, sf, st = engine.setRVec(wVec.size)
    assert rel_err(r2, rGpu) <= 1e-14       # setRVec as its own pass vs inside the kernels that complete dw: rounding of 1 / volRef
    assert abs(sf - (rf[:, :5] ** 2).sum()) <= 1e-9 * sf
    if nw > 5:
        assert abs(st - (rf[:, 5] ** 2).sum()) <= 1e-9 * max(st, 1e-300)
        g = engine.getRes(wVec.size).reshape(-1, nw)
        assert rel_err(g[:, 5] * prm.turbResScale, rr[:, 5]) <= 1e-14


def setup_block_with_bc(engine, dims, prm, spec, seed, **mk):
    """ONE block, six physical boundary faces, empty communication patterns: registered on the engine and committed
    to the reference's flowDoms(1,1,1).  Halos are made consistent with the boundary conditions on both sides."""
    from oracle import ref
    from adflow_amd.synth import make_bocos
    from adflow_amd.topology import CommPattern
    lvl = new_level(engine)
    prm = prm.replace(currentLevel=1, groundLevel=1)
    blk = make_block(*dims, prm, seed=seed, **mk)
    faces, nvisc = make_bocos(blk, prm, spec, seed=seed + 1)
    r = blk.copy()
    ref.alloc_doms(1, 1)
    ref.bind_block(r, prm)
    ref.set_bocos(faces, nvisc)
    ref.commit_block(1, 1)
    empty = CommPattern()
    for L in (1, 2):
        ref.set_internal_comm(1, L, empty)
    engine.set_options(prm)
    engine.register(blk, nn=1, level=lvl)
    engine.bc_register(faces, nvisc, nn=1, level=lvl)
    for L in (1, 2):
        engine.comm_register(lvl, L, empty)
    ref.call_level("applyAllBC", 1, 1)
    engine.applyAllBC(1, True)
    return blk, r, prm


def setup_blocks_with_bc(engine, prm, blocks_spec, seed=91, **mk):
    """Several unconnected blocks of different sizes on ONE level, each with its own set of boundary subfaces
    ({nn: (dims, spec, split)}; spec None = a block without subfaces, slot numbers may have gaps): what the
    level-batched boundary-condition launches see on a production multiblock mesh."""
    from oracle import ref
    from adflow_amd.synth import make_bocos
    from adflow_amd.topology import CommPattern
    engine.release_all()
    prm = prm.replace(currentLevel=1, groundLevel=1)
    blocks, bocos = {}, {}
    for nn, (dims, spec, split) in blocks_spec.items():
        blocks[nn] = make_block(*dims, prm, seed=seed + 13 * nn, **mk)
        if spec:
            bocos[nn] = make_bocos(blocks[nn], prm, spec, seed=seed + 13 * nn + 1, split=split)
    rblocks = {nn: b.copy() for nn, b in blocks.items()}
    ref.alloc_doms(max(blocks), 1)
    ref.bind_blocks(rblocks, prm, level=1, nlevels=1, alloc=False, bocos=bocos)
    empty = CommPattern()
    engine.set_options(prm)
    for nn, b in blocks.items():
        engine.register(b, nn=nn, level=1)
        if nn in bocos:
            engine.bc_register(*bocos[nn], nn=nn, level=1)
    for L in (1, 2):
        ref.set_internal_comm(1, L, empty)
        engine.comm_register(1, L, empty)
    return blocks, rblocks, prm


def check_pressure_early_exchange(engine, dims, prm, wall_face=5, seed=131, sweeps=1, topo=None, lattice_spec=None, **mk):
    """exchangePressureEarly (iteration.f90:44-53, smoothers.F90:363-365 / 674-676): two blocks joined in i, an inviscid wall with
    the normal-momentum pressure extrapolation spanning the interface.  bcEulerWall differentiates the pressure ALONG the wall,
    i.e. reads the first halo layer of the neighbouring block, and the reference refreshes exactly that layer (pressure only,
    whalo1) before applyAllBC in every stage.  The test first shows that the step depends on it (the reference WITHOUT the early
    exchange gives a different state), then compares the device against the reference's own smoother.
    topo + lattice_spec: a LatticeTopology instead (blocks of different size / orientation; the boundary kinds per outward lattice
    direction, the wall among them)."""
    from oracle import ref
    from adflow_amd.synth import make_bocos
    from adflow_amd.topology import BrickTopology, apply_local_copies_fast
    engine.release_all()
    prm = prm.replace(currentLevel=1, groundLevel=1, eulerWallBCTreatment=4, exchangePressureEarly=True)
    if topo is None:
        topo = BrickTopology(2, 1, 1, *dims, periodic=(False, False, False))
        others = {3: -6, 4: -6, 5: -6, 6: -6}
        others[wall_face] = -5
        specs = {1: {**others, 1: -6}, 2: {**others, 2: -6}}         # the i faces between the blocks stay 1-to-1
    else:
        lid = topo.local_ids()
        specs = {lid[g]: topo.boundary_spec(g, lattice_spec) for g in range(topo.nblocks)}
    name = "RungeKuttaSmoother" if prm.smoother == RungeKutta else "DADISmoother"

    def reference_run(early):
        blocks = make_brick(topo, prm, seed, **mk)
        pats = {L: topo.patterns(L)[0] for L in (1, 2)}
        apply_local_copies_fast(blocks, pats[2])
        bocos = {nn: make_bocos(blocks[nn], prm, specs[nn], seed=seed + nn) for nn in blocks}
        rblocks = {nn: b.copy() for nn, b in blocks.items()}
        ref.alloc_doms(max(rblocks), 1)
        ref.bind_blocks(rblocks, prm.replace(exchangePressureEarly=early), level=1, nlevels=1, alloc=False, bocos=bocos)
        for L in (1, 2):
            ref.set_internal_comm(1, L, pats[L])
        for nn in sorted(rblocks):
            ref.call_level("setPointers", 1, nn)
            ref.call("applyAllBC_block", 1)
        for s in range(sweeps):
            ref.load().ref_set_int(b"rkStage", 0)
            ref.call_level("timeStep", 1, 0)
            ref.call_level("initres", 1, 1, 5)
            ref.call_level("residual", 1)
            ref.call_level(name, 1)
        return blocks, rblocks, bocos, pats

    _, r_off, _, _ = reference_run(False)
    off = {nn: b["p"].copy() for nn, b in r_off.items()}
    blocks, rblocks, bocos, pats = reference_run(True)
    moved = max(rel_err(off[nn], rblocks[nn]["p"]) for nn in rblocks)
    assert moved > 1e-8, ("the early pressure exchange does not change this step", moved)
    engine.set_options(prm)
    for nn, b in blocks.items():
        engine.register(b, nn=nn, level=1)
        engine.bc_register(*bocos[nn], nn=nn, level=1)
    for L in (1, 2):
        engine.comm_register(1, L, pats[L])
    engine.applyAllBC(1, True)
    for s in range(sweeps):
        engine.timeStep(1, False)
        engine.residual(1, 0)
        getattr(engine, name)(1)
    assert_state(engine, blocks, rblocks, prm, f"{name} with exchangePressureEarly")
    return moved


def check_multiblock_bc(engine, prm, blocks_spec, seed=91, **mk):
    """applyAllBC over a level of several blocks with different subface lists, then (RANS) the SA solve and one
    smoother sweep on them: the device groups the subfaces of all blocks by kind / position in the block's list,
    the reference walks block by block."""
    from oracle import ref
    blocks, rblocks, prm = setup_blocks_with_bc(engine, prm, blocks_spec, seed, **mk)
    for nn in sorted(rblocks):
        ref.call_level("setPointers", 1, nn)
        ref.call("applyAllBC_block", 1)
    engine.applyAllBC(1, True)
    assert_state(engine, blocks, rblocks, prm, "multiblock applyAllBC")
    if prm.equations == RANSEquations:
        for it in range(prm.nSubIterTurb):
            for nn in sorted(rblocks):
                ref.call_level("setPointers", 1, nn)
                ref.call("sa_block", 0)
            ref.load().ref_call_level(b"whalo2_turb", 1, 6, 6)
        engine.turbSolveDDADI(1)
        assert_state(engine, blocks, rblocks, prm, "multiblock SA DDADI solve with BCs")
    name = "RungeKuttaSmoother" if prm.smoother == RungeKutta else "DADISmoother"
    ref.load().ref_set_int(b"rkStage", 0)
    ref.call_level("timeStep", 1, 0)
    ref.call_level("initres", 1, 1, 5)
    ref.call_level("residual", 1)
    ref.call_level(name, 1)
    engine.timeStep(1, False)
    engine.residual(1, 0)
    getattr(engine, name)(1)
    assert_state(engine, blocks, rblocks, prm, f"multiblock {name} with BCs")


def check_sa_solve_with_bc(engine, dims, prm, spec, seed=71, **mk):
    """turbSolveDDADI on a block with physical boundaries: bcTurbTreatment (bmt/bvt), their implicit part in the
    central jacobian (sa.F90:452-468, turbUtils.F90:986-1004) and applyAllTurbBCThisBlock(.true.) on the device."""
    from oracle import ref
    blk, r, prm = setup_block_with_bc(engine, dims, prm, spec, seed, **mk)
    for it in range(prm.nSubIterTurb):
        ref.call_level("setPointers", 1, 1)
        ref.call("sa_block", 0)
        ref.load().ref_call_level(b"whalo2_turb", 1, 6, 6)
    engine.turbSolveDDADI(1)
    assert_state(engine, {1: blk}, {1: r}, prm, f"SA DDADI solve with BCs {spec}")


def check_sa_solve(engine, topo, prm, seed=31, **mk):
    """turbSolveDDADI: nSubIterTurb x [sa_block(.false.) ; whalo2(nt1:nt2)]
    (turbAPI.F90:4-95, sa.F90:16-86,717-1268) on a periodic brick."""
    from oracle import ref
    blocks, rblocks = setup_brick(engine, topo, prm, seed, **mk)
    for it in range(prm.nSubIterTurb):
        for nn in sorted(rblocks):
            ref.call_level("setPointers", 1, nn)
            ref.call("sa_block", 0)
        ref.load().ref_call_level(b"whalo2_turb", 1, 6, 6)
    engine.turbSolveDDADI(1)
    assert_state(engine, blocks, rblocks, prm, "SA DDADI solve")
