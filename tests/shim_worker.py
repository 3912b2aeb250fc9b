"""Worker of the Fortran-shim execution tests: runs in a CLEAN process so that the C-ABI library (the HIP one on the MI355X,
or the tests/hostsim emulator build in the CPU-only CI) is loaded RTLD_GLOBAL BEFORE oracle/_ref/libadflow_ref.so.  The
reference library contains adflow_amd/fortran/adflow_gpu_shim.F90 compiled against the reference's own modules; its calls to
adflow_gpu_* then bind to the real library (symbol interposition over the abort stubs the oracle link generates).

  python shim_worker.py {hip|hostsim} {brick|bocos} equations spaceDiscr

brick : periodic 2x1x1 brick; ref_shim_roundtrip (Fortran: gpuRefreshOptions, gpuRegisterBlock, gpuRegisterComm from
        flowDoms / communication.F90, adflow_gpu_block_res with whalo2 on the device, residual back into flowDoms%dw) against the
        reference's own whalo2 + blockResCore on the same arrays.
bocos : one block with six physical boundary faces; additionally gpuRegisterBocos (BCData by c_loc) and the boundary
        conditions applied on the device inside block_res, against applyAllBC_block (+ turbulence BCs) + blockResCore."""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def main():
    kind, case, equations, sd = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    if kind == "hostsim":
        from hostsim.build import build
        path = build()
    else:
        from adflow_amd import build as B
        path = B.LIB
    ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)          # BEFORE the reference library
    from oracle import ref
    from adflow_amd import capi
    from adflow_amd.engine import Engine
    from adflow_amd.params import FlowParams, RANSEquations, DADI
    from adflow_amd.topology import BrickTopology, apply_local_copies_fast
    from adflow_amd.synth import make_block, make_bocos
    from util import rel_err, owned, TOL
    import checks

    lib = ref.load()
    lib.ref_shim_roundtrip.argtypes = [ctypes.c_int] * 4
    eng = Engine(0, _lib_path=path)                     # adflow_gpu_init; the Fortran side registers everything else
    prm = FlowParams(equations=equations, spaceDiscr=sd, vis4=0.1 if sd == 2 else 0.0156, smoother=DADI)
    turb = equations == RANSEquations
    nw = prm.nw
    rng = np.random.default_rng(3)
    flags = capi.RES_UPDATE_INTERMED | capi.RES_FLOW | (capi.RES_TURB if turb else 0) | capi.RES_HALO
    if case == "brick":
        topo = BrickTopology(2, 1, 1, 10, 6, 5)
        blocks = checks.make_brick(topo, prm, 11, stretch_k=2.0)
        pats = {L: topo.patterns(L)[0] for L in (1, 2)}
        apply_local_copies_fast(blocks, pats[2])
        ref.bind_blocks(blocks, prm)
        for L in (1, 2):
            ref.set_internal_comm(1, L, pats[L])
        # stale halos: the owned cells change after the halos were filled (E_t kept consistent with p)
        for b in blocks.values():
            s = (slice(2, b.il + 1), slice(2, b.jl + 1), slice(2, b.kl + 1))
            b["p"][s] *= rng.uniform(0.95, 1.05, b["p"][s].shape)
            w = b["w"]
            w[s + (0,)] *= rng.uniform(0.95, 1.05, w[s + (0,)].shape)
            w[..., 4] = b["p"] / (prm.gammaConstant - 1.0) + 0.5 * w[..., 0] * (w[..., 1] ** 2 + w[..., 2] ** 2 + w[..., 3] ** 2)
        ref._big_stack(lib.ref_shim_roundtrip, 1, flags, 0, 0)
        got = {nn: b["dw"].copy() for nn, b in blocks.items()}
        ref.call_level("whalo2", 1, 1, nw)
        for nn in sorted(blocks):
            ref.call_level("setPointers", 1, nn)
            ref.block_res_core(True, True, turb)
    else:
        blk = make_block(12, 8, 6, prm, seed=5, stretch_k=2.0)
        spec = {1: -3, 2: -4, 3: -1, 4: -4, 5: -1, 6: -6} if prm.viscous else {1: -3, 2: -4, 3: -2, 4: -4, 5: -2, 6: -6}
        faces, nvisc = make_bocos(blk, prm, spec, seed=6)
        blocks = {1: blk}
        ref.bind_blocks(blocks, prm, bocos={1: (faces, nvisc)})
        ref._big_stack(lib.ref_shim_roundtrip, 1, flags, 1, 0)
        got = {1: blk["dw"].copy()}
        ref.call_level("setPointers", 1, 1)
        if turb:
            ref.call("bcTurbTreatment")
            ref.call("applyAllTurbBCThisBlock", 1)
        ref.call("applyAllBC_block", 1)
        ref.block_res_core(True, True, turb)
    bad = 0
    for nn, b in blocks.items():
        for l in range(nw):
            e = rel_err(owned(b, got[nn][..., l]), owned(b, b["dw"][..., l]))
            if not e <= TOL:
                bad += 1
                print(f"block {nn} dw[{l}]: rel err {e:.3e}")
    eng.close()
    print("SHIM OK" if bad == 0 else "SHIM FAIL")
    sys.exit(0 if bad == 0 else 1)


if __name__ == "__main__":
    main()
