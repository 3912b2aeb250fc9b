"""GPU parity (real MI355X, through the C-ABI): laminar NS and RANS-SA residuals —
nodal gradients, viscous flux, SA source/advection/diffusion — against the
reference's own Fortran (oracle/_ref).  BASELINE configs 3-4 parity sizes."""
import pytest

import checks
from adflow_amd.params import (FlowParams, dissScalar, dissMatrix, upwind, EulerEquations, NSEquations, RANSEquations, secondOrder,
                               vorticity, noLimiter, vanAlbeda, minmod)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dims", [(16, 12, 8), (5, 3, 2)])
def test_laminar_ns(engine, dims):
    checks.check_block_res(engine, dims, FlowParams(equations=NSEquations), seed=4, stretch_k=2.0)


@pytest.mark.parametrize("sd", [dissScalar, dissMatrix, upwind])
def test_rans_sa_crm_parity_size(engine, sd):
    # BASELINE config 4 parity size: 24x20x10 blocks, 4a upwind / 4b matrix
    prm = FlowParams(equations=RANSEquations, spaceDiscr=sd, vis4=0.1 if sd == dissMatrix else 0.0156)
    checks.check_block_res(engine, (24, 20, 10), prm, seed=sd, stretch_k=3.0)


def test_rans_sa_tutorial_wing_parity_size(engine):
    # BASELINE config 3 parity size: 24 192 cells
    checks.check_block_res(engine, (48, 28, 18), FlowParams(equations=RANSEquations), seed=11, stretch_k=3.0)


def test_rans_sa_options(engine):
    prm = FlowParams(equations=RANSEquations, orderTurb=secondOrder, turbProd=vorticity, useQCR=True, useft2SA=False)
    checks.check_block_res(engine, (17, 9, 5), prm, seed=8, stretch_k=2.0)


def test_north_star_block_vs_reference_default_path(engine):
    """ONE block of BASELINE configs[3] at full size (160 x 128 x 64, RANS-SA, Roe upwind, van Albada) with the flags the bench
    times (updateIntermed = F) against the reference's default residual path blocketteResCore (blockette.F90:299-753)"""
    from adflow_amd.params import vanAlbeda
    prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=vanAlbeda)
    checks.check_block_res_vs_blockette(engine, (160, 128, 64), prm, False, seed=44, stretch_k=3.0)


def test_north_star_block_matrix_dissipation(engine):
    """config 4b at full size: one 160 x 128 x 64 block, RANS-SA with matrix dissipation (vis4 = 0.1), default flags, against
    blocketteResCore -- the size the bench quotes its 4b figure on"""
    prm = FlowParams(equations=RANSEquations, spaceDiscr=dissMatrix, vis4=0.1)
    checks.check_block_res_vs_blockette(engine, (160, 128, 64), prm, False, seed=45, stretch_k=3.0)


def test_multiblock_brick_at_bench_block_size(engine):
    """2 x 2 x 2 blocks of 64 x 48 x 32 cells (several tiles, k chunks and rounds per block; every block with face, edge and
    corner neighbours): stale halos -> whalo2 -> blocketteRes with the default flags, block by block against the reference"""
    from adflow_amd.topology import BrickTopology
    for sd in (upwind, dissMatrix):
        prm = FlowParams(equations=RANSEquations, spaceDiscr=sd, vis4=0.1 if sd == dissMatrix else 0.0156)
        checks.check_brick_block_res(engine, BrickTopology(2, 2, 2, 64, 48, 32), prm, seed=sd, stretch_k=2.0)


# the brick ends of the wall-bounded bench workload: viscous wall below (kMin), a symmetry plane (jMin), farfield elsewhere
WALL_BRICK = {1: -6, 2: -6, 3: -1, 4: -6, 5: -3, 6: -6}


def test_blockette_res_wall_bounded_brick(engine):
    """the reference's WHOLE blocketteRes (closures, turbulence + mean-flow boundary conditions, whalo2, core with storeWall) on
    wall-bounded bricks: dw of every block and viscSubface%tau / %q of every viscous subface; with and without the evaluation split
    around the exchange (which now also runs on meshes with walls)"""
    from adflow_amd.topology import BrickTopology
    prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
    open3 = (False, False, False)
    n = checks.check_blockette_res_with_bc(engine, BrickTopology(2, 1, 2, 10, 7, 6, periodic=open3), prm, WALL_BRICK, seed=3, stretch_k=2.0)
    assert n == 2
    # isothermal + adiabatic walls on several faces, matrix dissipation, QCR
    spec = {1: -4, 2: -6, 3: -3, 4: -15, 5: -3, 6: -9}
    prm2 = FlowParams(equations=RANSEquations, spaceDiscr=dissMatrix, vis4=0.1, useQCR=True)
    assert checks.check_blockette_res_with_bc(engine, BrickTopology(2, 2, 1, 9, 6, 5, periodic=open3), prm2, spec, seed=5, stretch_k=2.0) == 8
    # the split evaluation (tiles that read no halo cell first) on a mesh with walls
    checks.check_blockette_res_with_bc(engine, BrickTopology(2, 1, 1, 130, 11, 24, periodic=open3), prm.replace(muSuthDim=1.0), WALL_BRICK,
                                       seed=7, split_eval=2, stretch_k=2.0)
    # laminar NS, one block with six physical faces
    checks.check_blockette_res_with_bc(engine, BrickTopology(1, 1, 1, 12, 8, 6, periodic=open3), FlowParams(equations=NSEquations), spec, seed=9,
                                       stretch_k=2.0)
    # pressures at their floor: the energy whalo2 recomputes is written by the closures pass except in those cells
    checks.check_blockette_res_with_bc(engine, BrickTopology(2, 1, 2, 10, 7, 6, periodic=open3), FlowParams(equations=NSEquations), WALL_BRICK,
                                       seed=11, floor_p=True, stretch_k=2.0)


def test_north_star_block_with_six_physical_faces(engine):
    """ONE 160 x 128 x 64 block of BASELINE configs[3] whose six faces are physical boundaries (viscous wall, symmetry, farfield):
    the whole blocketteRes incl. viscSubface%tau / %q at full size -- what a near-body block of a CRM mesh executes"""
    from adflow_amd.topology import BrickTopology
    prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
    checks.check_blockette_res_with_bc(engine, BrickTopology(1, 1, 1, 160, 128, 64, periodic=(False, False, False)), prm, WALL_BRICK,
                                       seed=46, stretch_k=3.0)


def test_wall_bounded_brick_at_bench_block_size(engine):
    """2 x 2 x 2 blocks of 64 x 48 x 32 cells, the brick's ends physical boundaries (the layout of the bench's default workload):
    whole blocketteRes in one call, block by block against the reference, wall stress of the four lower blocks"""
    from adflow_amd.topology import BrickTopology
    prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
    n = checks.check_blockette_res_with_bc(engine, BrickTopology(2, 2, 2, 64, 48, 32, periodic=(False, False, False)), prm, WALL_BRICK,
                                           seed=47, stretch_k=2.0)
    assert n == 4


def test_evaluation_split_around_the_exchange(engine):
    """tuning split_eval = 2: whalo2 + blocketteRes in one call with the tiles that read no halo cell between the start and the end
    of the exchange, the others behind it: blocks with several interior tiles in every direction (interior / boundary partition of
    the SA march and of the chunk table of k_visc_gf), RANS and laminar, Roe and matrix dissipation; stale halos before the call"""
    from adflow_amd.topology import BrickTopology
    try:
        engine.set_tuning("split_eval", 2)
        engine.set_tuning("gf_cus", 2)            # small rounds: several k chunks per column, interior ones among them
        for prm in (FlowParams(equations=RANSEquations, spaceDiscr=upwind, muSuthDim=1.0),
                    FlowParams(equations=NSEquations, spaceDiscr=dissMatrix, vis4=0.1, muSuthDim=1.0)):
            checks.check_brick_block_res(engine, BrickTopology(2, 1, 1, 130, 14, 40), prm, seed=23, fused_halo=True, stretch_k=2.0)
        engine.set_tuning("gf_cus", 0)
        checks.check_brick_block_res(engine, BrickTopology(2, 2, 1, 9, 7, 5), FlowParams(equations=RANSEquations, spaceDiscr=upwind), seed=24,
                                     fused_halo=True, stretch_k=2.0)
    finally:
        engine.set_tuning("split_eval", 1)
        engine.set_tuning("gf_cus", 0)


def test_left_handed_block(engine):
    """a block whose (i, j, k) system is left-handed (mirror image): metric_block takes fact = -half.  The marching kernels that
    re-form the face normals from the nodes (SA, nodal gradients, time step) must do the same; update_geometry too."""
    prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
    checks.check_block_res(engine, (23, 9, 7), prm, seed=52, stretch_k=2.0, left_handed=True)
    checks.check_block_res(engine, (12, 6, 5), FlowParams(equations=RANSEquations), seed=53, stretch_k=2.0, left_handed=True)
    checks.check_block_res_vs_blockette(engine, (20, 9, 8), FlowParams(), False, seed=54, left_handed=True)
    checks.check_update_geometry(engine, (9, 8, 6), FlowParams(equations=NSEquations), {1: -6, 2: -6, 3: -3, 4: -6, 5: -1, 6: -1},
                                 stretch_k=2.0, left_handed=True)


def test_normals_not_derived_from_the_nodes_keep_the_stored_normals(engine):
    """the kernels re-form face normals from x only if the uploaded sI / sJ / sK ARE metric_block(x); a block whose normals were
    altered independently of its coordinates must be evaluated with the normals it was given, like the reference does"""
    from adflow_amd.synth import make_block
    prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
    blk = make_block(23, 9, 7, prm, seed=61, stretch_k=2.0)
    blk["sI"] *= 1.001
    blk["sK"] *= 0.9995
    checks.check_block_res(engine, (23, 9, 7), prm, blk=blk)
    blk = make_block(12, 6, 5, FlowParams(), seed=62)
    blk["sJ"] *= 1.002
    checks.check_block_res(engine, (12, 6, 5), FlowParams(), blk=blk)


def test_ns_rk_stage_residuals(engine):
    checks.check_rk_residual_sequence(engine, (12, 10, 6), FlowParams(equations=NSEquations), stretch_k=2.0)


def test_rans_full_size_block_vs_reference(engine):
    """BASELINE config 3 roofline-size block (128x128x96) against the reference itself."""
    checks.check_block_res(engine, (128, 128, 96), FlowParams(equations=RANSEquations), seed=5, stretch_k=3.0)


# ---- approximate residual of the preconditioner assembly (blocketteRes useDissApprox / useViscApprox) ----
@pytest.mark.parametrize("sd", [dissScalar, dissMatrix, upwind])
def test_block_res_approx(engine, sd):
    checks.check_block_res_approx(engine, (40, 12, 7), FlowParams(spaceDiscr=sd, sigma=0.2, vis4=0.05), visc_approx=False)
    checks.check_block_res_approx(engine, (24, 20, 10), FlowParams(equations=RANSEquations, spaceDiscr=sd, sigma=0.3, vis4=0.05),
                                  stretch_k=2.0)


def test_block_res_visc_approx_only(engine):
    from adflow_amd.params import NSEquations
    checks.check_block_res_approx(engine, (24, 20, 10), FlowParams(equations=NSEquations, sigma=0.3), diss_approx=False,
                                  visc_approx=True, stretch_k=2.0)


def test_wall_stress_storage(engine):
    """a7 / a17 useStoreWall: viscSubface%tau, %q on all six block faces, split subfaces, QCR"""
    from adflow_amd.params import NSEquations
    checks.check_wall_stress(engine, (70, 9, 8), FlowParams(equations=NSEquations), {1: -3, 2: -4, 3: -3, 4: -6, 5: -4, 6: -3},
                             stretch_k=2.0)
    checks.check_wall_stress(engine, (24, 10, 8), FlowParams(equations=RANSEquations, useQCR=True),
                             {1: -6, 2: -6, 3: -1, 4: -4, 5: -3, 6: -6}, split={5: -6, 4: -3}, stretch_k=2.0)


def test_inviscid_march_variants(engine):
    """tuning inviscid_march: 0 = cell-gather kernel for matrix / upwind, 1 = gather kernel for NS / RANS scalar JST (2, the default: marching form there too)"""
    from adflow_amd.params import NSEquations
    try:
        engine.set_tuning("inviscid_march", 0)
        for sd in (dissMatrix, upwind):
            checks.check_block_res(engine, (70, 9, 8), FlowParams(equations=RANSEquations, spaceDiscr=sd), seed=sd, stretch_k=2.0)
        engine.set_tuning("inviscid_march", 1)
        engine.set_tuning("march_kch", 5)
        checks.check_block_res(engine, (70, 9, 12), FlowParams(equations=RANSEquations), seed=6, stretch_k=2.0)
        checks.check_rk_residual_sequence(engine, (24, 10, 8), FlowParams(equations=NSEquations), stretch_k=2.0)
        engine.set_tuning("inviscid_march", 2)          # the default form again, with partial k chunks
        checks.check_block_res(engine, (70, 9, 12), FlowParams(equations=RANSEquations), seed=7, stretch_k=2.0)
        checks.check_rk_residual_sequence(engine, (24, 10, 8), FlowParams(equations=NSEquations), stretch_k=2.0)
    finally:
        engine.set_tuning("inviscid_march", 2)
        engine.set_tuning("march_kch", 32)


# ---- round 2: the synthetic states have a tiny viscous part (Re ~ 1e7: 1e-6 of dw); with muSuthDim = 1 the viscous flux is as
#      large as the inviscid one, so an error in the nodal gradients / face fluxes cannot hide below the 1e-10 bar
@pytest.mark.parametrize("eq,qcr", [(NSEquations, False), (RANSEquations, False), (RANSEquations, True)])
def test_viscous_dominated(engine, eq, qcr):
    for sd in (upwind, dissScalar):
        prm = FlowParams(equations=eq, spaceDiscr=sd, useQCR=qcr, muSuthDim=1.0)
        checks.check_block_res(engine, (70, 9, 12), prm, seed=21, stretch_k=2.0)
    checks.check_rk_residual_sequence(engine, (24, 10, 8), FlowParams(equations=NSEquations, muSuthDim=1.0), stretch_k=2.0)


def test_viscous_kernel_variants(engine):
    """tuning viscous_tiled: 2 = k-marching gradient + face kernels (default), 0 = gather pair;
    roe_march: 0 = per-face reconstruction kernel; partial tiles in i, j and the k chunk; blanked cells"""
    try:
        for vt in (2, 0):
            engine.set_tuning("viscous_tiled", vt)
            prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind, muSuthDim=1.0)
            checks.check_block_res(engine, (63, 6, 35), prm, seed=vt, stretch_k=2.0, holes=0.05)
        engine.set_tuning("viscous_tiled", 2)
        engine.set_tuning("roe_march", 0)
        checks.check_block_res(engine, (63, 6, 9), FlowParams(equations=RANSEquations, spaceDiscr=upwind), seed=3, stretch_k=2.0)
        engine.set_tuning("roe_march", 1)
        for sm in (0,):             # the gather kernel instead of the SA march
            engine.set_tuning("sa_march", sm)
            prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind, orderTurb=secondOrder)
            checks.check_block_res(engine, (63, 11, 35), prm, seed=10 + sm, stretch_k=2.0, holes=0.05)
        engine.set_tuning("sa_march", 1)
        # the viscous march in front of each inviscid march over the tile table (Roe, matrix dissipation, scalar JST), QCR, minmod
        for prm in (FlowParams(equations=RANSEquations, spaceDiscr=upwind, useQCR=True, muSuthDim=1.0),
                    FlowParams(equations=NSEquations, spaceDiscr=upwind, limiter=minmod, muSuthDim=1.0),
                    FlowParams(equations=RANSEquations, spaceDiscr=dissMatrix, vis4=0.1, muSuthDim=1.0),
                    FlowParams(equations=NSEquations, spaceDiscr=dissScalar, muSuthDim=1.0)):
            checks.check_block_res(engine, (63, 6, 35), prm, seed=67, stretch_k=2.0, holes=0.05)
        for im in (0, 2):            # the gather inviscid kernel: the viscous march then completes dw itself
            engine.set_tuning("inviscid_march", im)
            checks.check_block_res(engine, (63, 6, 35), FlowParams(equations=RANSEquations, spaceDiscr=dissMatrix, vis4=0.1, muSuthDim=1.0),
                                   seed=68, stretch_k=2.0, holes=0.05)
        for xt, cus in ((0, 0), (1, -1), (2, 3)):     # chunk tables of the SA / fused viscous marches: tiles in launch order; chunks of
                                                      # march_kch planes without the round fit; rounds of six workgroups
            engine.set_tuning("xcd_tiles", xt)
            engine.set_tuning("gf_cus", cus)
            prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind, orderTurb=secondOrder)
            checks.check_block_res(engine, (63, 11, 35), prm, seed=50 + cus, stretch_k=2.0, holes=0.05)
        engine.set_tuning("xcd_tiles", 2)
        engine.set_tuning("gf_cus", 0)
        for mx in (0, 1):           # face normals from the arrays everywhere / re-formed from the nodes in the SA march only;
                                    # the default 5 = SA march + time step
            engine.set_tuning("metric_from_x", mx)
            prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
            checks.check_block_res(engine, (63, 11, 35), prm, seed=20 + mx, stretch_k=2.0, holes=0.05)
    finally:
        engine.set_tuning("viscous_tiled", 2)
        engine.set_tuning("roe_march", 1)
        engine.set_tuning("sa_march", 1)
        engine.set_tuning("xcd_tiles", 2)
        engine.set_tuning("gf_cus", 0)
        engine.set_tuning("inviscid_march", 2)
        engine.set_tuning("metric_from_x", 5)


def test_visc_gradient_fused(engine):
    """k_visc_gf: nodal gradients and viscous fluxes in one kernel, the gradients stay in an LDS ring.
    Partial tiles in i (60 columns) / j (3 rows) / the k chunk, blanked cells, QCR, laminar NS, matrix / scalar dissipation, the
    kernel completing dw itself (persistent fw of the RK stages), the stored-gradient variant (wall stress, updateIntermed), k chunks
    of march_kch planes (gf_cus = -1) and chunks fitted to rounds of 2 x CUs workgroups as on a device with 1 / 3 CUs."""
    try:
        prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind, muSuthDim=1.0)
        checks.check_block_res(engine, (63, 7, 35), prm, seed=5, stretch_k=2.0, holes=0.05)
        checks.check_block_res(engine, (61, 9, 33), prm.replace(useQCR=True), seed=6, stretch_k=2.0)
        checks.check_block_res(engine, (7, 5, 3), FlowParams(equations=NSEquations), seed=7, stretch_k=2.0)
        checks.check_block_res(engine, (124, 13, 5), FlowParams(equations=NSEquations, spaceDiscr=dissMatrix, muSuthDim=1.0), seed=8, stretch_k=2.0)
        checks.check_block_res(engine, (1, 1, 1), FlowParams(equations=NSEquations, muSuthDim=1.0), seed=9)
        checks.check_rk_residual_sequence(engine, (12, 10, 6), FlowParams(equations=NSEquations, muSuthDim=1.0), stretch_k=2.0)
        checks.check_wall_stress(engine, (9, 8, 7), FlowParams(equations=RANSEquations), {1: -6, 2: -6, 3: -3, 4: -4, 5: -3, 6: -6},
                                 stretch_k=2.0)
        engine.set_tuning("gf_cus", -1)           # k chunks of march_kch planes
        engine.set_tuning("march_kch", 5)
        checks.check_block_res(engine, (20, 4, 13), prm, seed=11, stretch_k=2.0)
        engine.set_tuning("march_kch", 32)
        for cus in (1, 3, 5, 27):      # rounds of 2, 6, 10, 54 workgroups: 10 and 54 are not multiples of 8 (round-3 advisor finding)
            engine.set_tuning("gf_cus", cus)
            checks.check_block_res(engine, (70, 7, 21), prm, seed=12 + cus, stretch_k=2.0, holes=0.05)
    finally:
        engine.set_tuning("march_kch", 32)
        engine.set_tuning("gf_cus", 0)


def test_block_res_without_intermediates(engine):
    """default flags of blocketteRes (updateIntermed = F): dw only; the spectral radii are not an output"""
    from oracle import ref
    from util import owned, rel_err, TOL
    from adflow_amd.synth import make_block
    for sd in (upwind, dissMatrix, dissScalar):
        prm = FlowParams(equations=RANSEquations, spaceDiscr=sd, vis4=0.1 if sd == dissMatrix else 0.0156)
        engine.release_all()
        blk = make_block(24, 10, 8, prm, seed=sd, stretch_k=2.0)
        r = checks.ref_bind(blk, prm)
        ref.block_res_core(False, True, True)
        engine.set_options(prm)
        engine.register(blk)
        engine.blocketteRes(1, False, True, True)
        dw = engine.download_residual()
        for l in range(blk.nw):
            assert rel_err(owned(blk, dw[..., l]), owned(blk, r["dw"][..., l])) <= TOL, (sd, l)


@pytest.mark.parametrize("sd", [dissScalar, dissMatrix, upwind])
@pytest.mark.parametrize("update", [False, True])
def test_block_res_vs_blockette_core(engine, sd, update):
    """the reference's default residual path (blocketteResCore) on the config 3 / 4 parity sizes"""
    prm = FlowParams(equations=RANSEquations, spaceDiscr=sd, vis4=0.1 if sd == dissMatrix else 0.0156)
    checks.check_block_res_vs_blockette(engine, (24, 20, 10), prm, update, seed=sd, stretch_k=3.0)
    checks.check_block_res_vs_blockette(engine, (17, 9, 11), FlowParams(spaceDiscr=sd, vis4=0.1 if sd == dissMatrix else 0.0156), update, seed=sd)


def test_foreign_normals_then_own_nodes(engine):
    """round-4 advisor: an upload of sI from a foreign buffer followed by x from the registered pointer must leave the kernels on the
    stored normals"""
    checks.check_foreign_normals_then_own_nodes(engine, (63, 11, 35), FlowParams(equations=RANSEquations, spaceDiscr=upwind), stretch_k=2.0)


def test_blockette_res_with_bc_on_thin_blocks(engine):
    """the whole blocketteRes with boundary subfaces on blocks that are thin in one or several directions; floored pressures"""
    from adflow_amd.topology import BrickTopology
    spec = {1: -6, 2: -6, 3: -1, 4: -6, 5: -3, 6: -6}
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
    for dims in ((70, 9, 11), (130, 5, 3), (3, 2, 1), (24, 16, 2)):
        checks.check_blockette_res_with_bc(engine, BrickTopology(2, 1, 1, *dims, periodic=(False, False, False)), rans, spec, stretch_k=2.0)
    checks.check_blockette_res_with_bc(engine, BrickTopology(2, 2, 2, 64, 48, 32, periodic=(False, False, False)), rans, spec, floor_p=True,
                                       stretch_k=2.0)


def test_split_evaluation_error_exit_joins_the_side_queue(engine):
    """round-4 verdict, weak 13: an error behind the fork of the split evaluation (tuning test_fault bit 1) is reported and the side
    queue joined: the very next evaluation -- split again -- is correct"""
    from adflow_amd.topology import BrickTopology
    spec = {1: -6, 2: -6, 3: -1, 4: -6, 5: -3, 6: -6}
    rans = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
    topo = BrickTopology(2, 1, 1, 70, 9, 11, periodic=(False, False, False))
    try:
        engine.set_tuning("test_fault", 2)
        with pytest.raises(Exception, match="test_fault"):
            checks.check_blockette_res_with_bc(engine, topo, rans, spec, split_eval=2, stretch_k=2.0)
    finally:
        engine.set_tuning("test_fault", 0)
        engine.set_tuning("split_eval", 1)
    checks.check_blockette_res_with_bc(engine, topo, rans, spec, split_eval=2, stretch_k=2.0)


def test_wall_bounded_brick_through_rccl_self_and_split(engine):
    """round-4 verdict, next 8 (i): the product's own exchange on the wall-bounded 2 x 2 x 2 brick with every interface as an RCCL
    message to the own rank (comm_self) and the evaluation split around it (the default when the pattern has messages): derived values,
    boundary conditions on the main queue while the interior tiles of the SA and viscous marches run on the side queue, pack ->
    ncclSend / ncclRecv -> unpack, the boundary tiles, the inviscid march; 64 x 48 x 32 and 128 x 64 x 32 blocks"""
    from adflow_amd.topology import BrickTopology
    prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind)
    engine.comm_init_single()
    try:
        engine.set_tuning("comm_self", 1)
        for dims, seed in (((64, 48, 32), 47), ((128, 64, 32), 53)):
            n = checks.check_blockette_res_with_bc(engine, BrickTopology(2, 2, 2, *dims, periodic=(False, False, False)), prm, WALL_BRICK,
                                                   seed=seed, stretch_k=2.0)
            assert n == 4
    finally:
        engine.set_tuning("comm_self", 0)
