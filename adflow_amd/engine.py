"""Host-side mirror of the reference's operator surface for the hot path.

Method names follow the reference's shell routines:
`timeStep`, `initres`, `residual`, `blocketteRes`, `RungeKuttaSmoother`,
`DADISmoother`, `whalo2`; each forwards to the C-ABI of include/adflow_gpu.h.
Blocks are addressed like `flowDoms(nn, level, sps)` (1-based).
"""
from __future__ import annotations

import ctypes
from typing import Dict, Tuple

import numpy as np

from . import capi
from .params import FlowParams


class Engine:
    def __init__(self, device: int = 0, _lib_path=None):
        self.lib = capi.load(_lib_path)
        self._chk(self.lib.adflow_gpu_init(device))
        self.blocks: Dict[Tuple[int, int, int], object] = {}
        self._descs = {}
        self.prm = None

    def _chk(self, rc):
        capi.check(rc, self.lib)

    # ---- lifetime ---------------------------------------------------------
    def close(self):
        if self.lib is not None:
            self.lib.adflow_gpu_finalize()
            self.lib = None

    def device_name(self) -> str:
        buf = ctypes.create_string_buffer(256)
        self._chk(self.lib.adflow_gpu_device_name(buf, 256))
        return buf.value.decode()

    # ---- data model -------------------------------------------------------
    def set_options(self, prm: FlowParams):
        self.prm = prm
        o = capi.opts_from_params(prm)
        self._chk(self.lib.adflow_gpu_set_options(ctypes.byref(o)))

    def register(self, blk, nn: int = 1, level: int = 1, sps: int = 1, upload: bool = True):
        """flowDoms(nn,level,sps) <- blk ; host arrays stay owned by `blk`."""
        a = blk.a
        ib, jb, kb, ie, je, ke = blk.ib, blk.jb, blk.kb, blk.ie, blk.je, blk.ke
        for name, shape in (("dw", (ib + 1, jb + 1, kb + 1, blk.nw)), ("fw", (ib + 1, jb + 1, kb + 1, 5)),
                            ("dtl", (ie, je, ke)), ("radI", (ie, je, ke)), ("radJ", (ie, je, ke)),
                            ("radK", (ie, je, ke))):
            if name not in a:
                a[name] = np.zeros(shape, order="F")
        d = capi.desc_from_block(blk)
        self._descs[(nn, level, sps)] = d
        self.blocks[(nn, level, sps)] = blk
        self._chk(self.lib.adflow_gpu_block_register(nn, level, sps, ctypes.byref(d)))
        if upload:
            self.upload_geometry(nn, level, sps)
            self.upload_state(nn, level, sps)

    def release(self, nn=1, level=1, sps=1):
        self._chk(self.lib.adflow_gpu_block_release(nn, level, sps))
        self.blocks.pop((nn, level, sps), None)
        self._descs.pop((nn, level, sps), None)

    def release_all(self):
        self._chk(self.lib.adflow_gpu_release_all())
        self.blocks.clear()
        self._descs.clear()

    def upload_geometry(self, nn=1, level=1, sps=1):
        self._chk(self.lib.adflow_gpu_upload_geometry(nn, level, sps))

    def upload_state(self, nn=1, level=1, sps=1):
        self._chk(self.lib.adflow_gpu_upload_state(nn, level, sps))

    def download_state(self, nn=1, level=1, sps=1):
        self._chk(self.lib.adflow_gpu_download_state(nn, level, sps))

    def download_residual(self, nn=1, level=1, sps=1):
        self._chk(self.lib.adflow_gpu_download_residual(nn, level, sps))
        return self.blocks[(nn, level, sps)]["dw"]

    def download_array(self, which: int, out: np.ndarray, nn=1, level=1, sps=1):
        assert out.flags["F_CONTIGUOUS"] and out.dtype == np.float64
        self._chk(self.lib.adflow_gpu_download_array(nn, level, sps, which, out.ctypes.data))
        return out

    def upload_array(self, which: int, src: np.ndarray, nn=1, level=1, sps=1):
        assert src.flags["F_CONTIGUOUS"] and src.dtype == np.float64
        self._chk(self.lib.adflow_gpu_upload_array(nn, level, sps, which, src.ctypes.data))

    # ---- the hot path (reference shell-routine names) -----------------------
    def timeStep(self, level=1, onlyRadii=False):
        self._chk(self.lib.adflow_gpu_time_step(level, int(onlyRadii)))

    def initres(self, level, varStart, varEnd):
        self._chk(self.lib.adflow_gpu_initres(level, varStart, varEnd))

    def residual(self, level=1, rkStage=0):
        self._chk(self.lib.adflow_gpu_residual(level, rkStage))

    def referenceShockSensor(self, level=1):
        self._chk(self.lib.adflow_gpu_reference_shock_sensor(level))

    def registerWallAssociation(self, surfNodeIndices: np.ndarray, uv: np.ndarray, nn=1, level=1, sps=1):
        """flowDoms%surfNodeIndices (4,nx,ny,nz) int32 / %uv (2,nx,ny,nz), Fortran order"""
        assert surfNodeIndices.dtype == np.int32 and surfNodeIndices.flags["F_CONTIGUOUS"] and uv.flags["F_CONTIGUOUS"]
        self._chk(self.lib.adflow_gpu_wall_distance_register(nn, level, sps, surfNodeIndices.ctypes.data, uv.ctypes.data))

    def updateWallDistancesQuickly(self, xSurf: np.ndarray, level=1):
        xSurf = np.ascontiguousarray(xSurf, dtype=np.float64)
        self._chk(self.lib.adflow_gpu_update_wall_distances(level, xSurf.ctypes.data, xSurf.size))

    def setupStateResidualMatrix(self, level=1, usePC=True, frozenTurb=False, useTurbOnly=False, viscPC=False, delta=1e-9, useAD=False):
        """adjointUtils::setupStateResidualMatrix (adjointUtils.F90:7-715) without the PETSc calls: the stencil blocks stay on the
        device; jacobianBlocks() brings one block's over.  useAD = False: coloured finite differences with step delta; True: one
        forward-mode (dual-number) evaluation per colour and state variable, the exact derivative (adjointUtils.F90:227-409)."""
        flags = (capi.JAC_PC if usePC else 0) | (capi.JAC_FROZEN_TURB if frozenTurb else 0) \
            | (capi.JAC_TURB_ONLY if useTurbOnly else 0) | (capi.JAC_VISC_PC if viscPC else 0) | (capi.JAC_USE_AD if useAD else 0)
        self._chk(self.lib.adflow_gpu_fd_jacobian(level, flags, float(delta)))

    def releaseWorkspace(self) -> int:
        """gives the dual-number slab a forward-mode assembly keeps between calls back to the device; returns the bytes released"""
        n = ctypes.c_int64(0)
        self._chk(self.lib.adflow_gpu_release_workspace(ctypes.byref(n)))
        return int(n.value)

    def selftestMath(self, which: int, x, a=None):
        """the kernels' fast division / root / power forms on the arguments x (and exponents / numerators a): returns
        (plain value, dual value, dual derivative) -- csrc/internal.h, csrc/kernels_ad.hip"""
        x = np.ascontiguousarray(x, dtype=np.float64)
        a = np.ones_like(x) if a is None else np.ascontiguousarray(np.broadcast_to(a, x.shape), dtype=np.float64)
        y = np.zeros_like(x)
        dy = np.zeros((x.size, 2))
        self._chk(self.lib.adflow_gpu_selftest_math(int(which), x.ctypes.data, a.ctypes.data, x.size, y.ctypes.data, dy.ctypes.data))
        return y, dy[:, 0].reshape(x.shape), dy[:, 1].reshape(x.shape)

    def jacobianInfo(self):
        ns, nst = ctypes.c_int32(), ctypes.c_int32()
        self._chk(self.lib.adflow_gpu_jacobian_info(ctypes.byref(ns), ctypes.byref(nst), None))
        st = np.zeros((nst.value, 3), dtype=np.int32, order="F")
        self._chk(self.lib.adflow_gpu_jacobian_info(None, None, st.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
        return ns.value, st

    def jacobianBlocks(self, nn=1, level=1, sps=1):
        """(nx, ny, nz, nState, nState, nStencil): blk(ll, l) of stencil entry s at every owned row cell"""
        ns, st = self.jacobianInfo()
        blk = self.blocks[(nn, level, sps)]
        out = np.zeros((blk.nx, blk.ny, blk.nz, ns, ns, st.shape[0]), order="F")
        self._chk(self.lib.adflow_gpu_download_jacobian(nn, level, sps, out.ctypes.data))
        return out

    def jacobianRows(self, nn=1, level=1, sps=1):
        """(nState, nState, nStencil, nx, ny, nz): the same blocks, contiguous per row cell (one MatSetValuesBlocked per cell and
        stencil entry reads nState^2 consecutive doubles)"""
        ns, st = self.jacobianInfo()
        blk = self.blocks[(nn, level, sps)]
        out = np.zeros((ns, ns, st.shape[0], blk.nx, blk.ny, blk.nz), order="F")
        self._chk(self.lib.adflow_gpu_download_jacobian_rows(nn, level, sps, out.ctypes.data))
        return out

    def blocketteRes(self, level=1, updateIntermed=True, flowRes=True, turbRes=True, dissApprox=False, viscApprox=False,
                     useBlockettes=False, halo=False, closures=False):
        """halo: also the part of blocketteRes in front of the core -- boundary conditions and whalo2 (ADFLOW_RES_HALO);
        closures: and the derived values in front of those -- computePressureSimple, computeLamViscosity, computeEddyViscosity
        (ADFLOW_RES_CLOSURES, blockette.F90:199-203).  Both = the reference's whole blocketteRes."""
        flags = (capi.RES_UPDATE_INTERMED if updateIntermed else 0) | (capi.RES_FLOW if flowRes else 0) \
            | (capi.RES_TURB if turbRes else 0) | (32 if dissApprox else 0) | (64 if viscApprox else 0) \
            | (128 if useBlockettes else 0) | (capi.RES_HALO if halo else 0) | (capi.RES_CLOSURES if closures else 0)
        self._chk(self.lib.adflow_gpu_block_res(level, flags))

    def bc_register(self, faces, nViscBocos: int = 0, nn: int = 1, level: int = 1, sps: int = 1):
        """flowDoms(nn,level,sps)%BCType/BCFaceID/BCData -> device.  `faces`: list of dicts with bcType, faceID,
        icBeg, icEnd, jcBeg, jcEnd and the BCData members (Fortran-order float64 arrays) the kind needs."""
        arr = (capi.AdflowBcSubface * max(len(faces), 1))()
        for m, f in enumerate(faces):
            for k in ("bcType", "faceID", "icBeg", "icEnd", "jcBeg", "jcEnd"):
                setattr(arr[m], k, int(f[k]))
            arr[m].subsonicInletTreatment = int(f.get("subsonicInletTreatment", 0))
            if f.get("symNorm") is not None:
                for q in range(3):
                    arr[m].symNorm[q] = float(f["symNorm"][q])
            for k in capi.BC_ARRAYS:
                a = f.get(k)
                if a is not None:
                    assert a.flags["F_CONTIGUOUS"] and a.dtype == np.float64, k
                    setattr(arr[m], k, a.ctypes.data)
        self._chk(self.lib.adflow_gpu_bc_register(nn, level, sps, len(faces), int(nViscBocos), arr))

    def wall_stress(self, shape, mm: int, nn: int = 1, level: int = 1, sps: int = 1):
        """viscSubface(mm)%tau, %q of viscous subface mm (1-based) over its owned face cells `shape` = (n1, n2)"""
        tau = np.zeros(tuple(shape) + (6,), order="F")
        q = np.zeros(tuple(shape) + (3,), order="F")
        self._chk(self.lib.adflow_gpu_download_wall_stress(nn, level, sps, mm, tau.ctypes.data, q.ctypes.data))
        return tau, q

    def upload_coordinates(self, nn=1, level=1, sps=1):
        self._chk(self.lib.adflow_gpu_upload_coordinates(nn, level, sps))

    def update_geometry(self, level=1):
        """volume_block + metric_block + boundaryNormals on the device"""
        self._chk(self.lib.adflow_gpu_update_geometry(level))

    def comm_register_periodic(self, level, nLayers, periodic):
        """periodic: list of dicts rotMatrix (3,3), rotCenter (3), translation (3), block (n) int32, indices (n,3) int32 F-order"""
        arr = (capi.AdflowPeriodicData * max(len(periodic), 1))()
        keep = []
        for m, pd in enumerate(periodic):
            R = np.asfortranarray(pd["rotMatrix"], dtype=np.float64)
            for q in range(9):
                arr[m].rotMatrix[q] = float(R.ravel(order="F")[q])
            for q in range(3):
                arr[m].rotCenter[q] = float(pd["rotCenter"][q])
                arr[m].translation[q] = float(pd["translation"][q])
            blk = np.ascontiguousarray(pd["block"], np.int32)
            idx = np.asfortranarray(pd["indices"], np.int32)
            keep += [blk, idx]
            arr[m].nHalos = int(blk.size)
            arr[m].block = blk.ctypes.data
            arr[m].indices = idx.ctypes.data
        self._chk(self.lib.adflow_gpu_comm_register_periodic(level, nLayers, len(periodic), arr))

    def actuator_register(self, regions):
        """regions: list of dicts block (n) int32, cellIDs (3,n) int32 F-order, force (3), heat, volume, relaxStart, relaxEnd"""
        arr = (capi.AdflowActuatorRegion * max(len(regions), 1))()
        keep = []
        for m, r in enumerate(regions):
            blk = np.ascontiguousarray(r["block"], np.int32)
            ids = np.asfortranarray(r["cellIDs"], np.int32)
            keep += [blk, ids]
            arr[m].nCellIDs = int(blk.size)
            arr[m].block, arr[m].cellIDs = blk.ctypes.data, ids.ctypes.data
            for q in range(3):
                arr[m].force[q] = float(r["force"][q])
            arr[m].heat, arr[m].volume = float(r["heat"]), float(r["volume"])
            arr[m].relaxStart, arr[m].relaxEnd = float(r.get("relaxStart", -1.0)), float(r.get("relaxEnd", -1.0))
        self._chk(self.lib.adflow_gpu_actuator_register(len(regions), arr))

    def xhalo(self, level=1):
        """xhalo_block of every block of the level"""
        self._chk(self.lib.adflow_gpu_xhalo(level))

    def coarseOwnedCoordinates(self, coarseLevel):
        self._chk(self.lib.adflow_gpu_coarse_coordinates(coarseLevel))

    def exchangeCoor(self, level=1):
        self._chk(self.lib.adflow_gpu_exchange_coor(level))

    def applyAllBC(self, level=1, secondHalo=True):
        self._chk(self.lib.adflow_gpu_apply_all_bc(level, int(secondHalo)))

    def set_tuning(self, key: str, value: int):
        self._chk(self.lib.adflow_gpu_set_tuning(key.encode(), int(value)))

    def set_async(self, on: bool):
        """Entry points only enqueue on the library stream; order with sync()."""
        self._chk(self.lib.adflow_gpu_set_async(int(on)))

    def RungeKuttaSmoother(self, level=1):
        self._chk(self.lib.adflow_gpu_rk_smooth(level))

    def DADISmoother(self, level=1):
        self._chk(self.lib.adflow_gpu_dadi_smooth(level))

    def turbSolveDDADI(self, level=1):
        self._chk(self.lib.adflow_gpu_sa_solve(level))

    # ---- Newton-Krylov glue (nksolver.* of src/f2py/adflow.pyf:394-421) -----
    def setW(self, wVec: np.ndarray):
        assert wVec.dtype == np.float64 and wVec.flags["C_CONTIGUOUS"]
        self._chk(self.lib.adflow_gpu_set_w_vec(wVec.ctypes.data, wVec.size))

    def setRVec(self, n: int):
        """-> (rVec, sum flow^2, sum turb^2)"""
        r = np.zeros(n)
        s2 = np.zeros(2)
        self._chk(self.lib.adflow_gpu_get_r_vec(r.ctypes.data, n, s2.ctypes.data))
        return r, s2[0], s2[1]

    def getRes(self, n: int):
        r = np.zeros(n)
        self._chk(self.lib.adflow_gpu_get_res(r.ctypes.data, n))
        return r

    def FormFunction_mf(self, wVec: np.ndarray):
        """setW + blocketteRes + setRVec (NKSolvers.F90:437-461)."""
        assert wVec.dtype == np.float64 and wVec.flags["C_CONTIGUOUS"]
        r = np.zeros_like(wVec)
        self._chk(self.lib.adflow_gpu_nk_residual(wVec.ctypes.data, r.ctypes.data, wVec.size))
        return r

    # ---- multigrid ----------------------------------------------------------
    def transferToCoarseGrid(self, level=1):
        self._chk(self.lib.adflow_gpu_transfer_to_coarse(level))

    def transferToFineGrid(self, level=1):
        self._chk(self.lib.adflow_gpu_transfer_to_fine(level))

    def executeMGCycle(self, cycling):
        c = np.ascontiguousarray(cycling, np.int32)
        self._chk(self.lib.adflow_gpu_mg_cycle(c.ctypes.data, c.size))

    # ---- halo exchange ------------------------------------------------------
    def comm_register(self, level: int, nLayers: int, cp):
        """commPatternCell_{1st,2nd}(level) + internalCell_{1st,2nd}(level)."""
        c = capi.comm_pattern_struct(cp)
        self._chk(self.lib.adflow_gpu_comm_register(level, nLayers, ctypes.byref(c)))

    def comm_init_single(self):
        """RCCL communicator of ONE rank (adflow_gpu_comm_unique_id + adflow_gpu_comm_init(0, 1, id)): what a multi-rank host does
        with the id broadcast over MPI / torch.distributed; enough for messages to the own rank (tuning comm_self)."""
        if getattr(self, "_comm_ready", False):
            return
        raw = (ctypes.c_char * 128)()
        self._chk(self.lib.adflow_gpu_comm_unique_id(raw))
        self._chk(self.lib.adflow_gpu_comm_init(0, 1, raw))
        self._comm_ready = True

    def comm_info(self):
        """(rank, nranks as given to adflow_gpu_comm_init; ncclCommCount, ncclCommUserRank as the communicator reports them, -1 before)"""
        v = [ctypes.c_int() for _ in range(4)]
        self._chk(self.lib.adflow_gpu_comm_info(*[ctypes.byref(x) for x in v]))
        return tuple(int(x.value) for x in v)

    def whalo1(self, level, start, end, commPressure=True, commGamma=True, commViscous=True):
        self._chk(self.lib.adflow_gpu_halo_exchange(level, start, end, int(commPressure), int(commViscous), 1))

    def whalo2(self, level, start, end, commPressure=True, commGamma=True, commViscous=True):
        self._chk(self.lib.adflow_gpu_halo_exchange(level, start, end, int(commPressure), int(commViscous), 2))

    def res_norms(self, level=1, n=5):
        out = np.zeros(n)
        self._chk(self.lib.adflow_gpu_res_norms(level, out.ctypes.data, n))
        return out

    # ---- instrumentation ----------------------------------------------------
    def event_record(self, slot: int):
        self._chk(self.lib.adflow_gpu_event_record(slot))

    def event_elapsed_ms(self, a: int, b: int) -> float:
        ms = ctypes.c_double()
        self._chk(self.lib.adflow_gpu_event_elapsed_ms(a, b, ctypes.byref(ms)))
        return ms.value

    def march_stats(self, level: int = 1):
        out = (ctypes.c_double * 4)()
        self._chk(self.lib.adflow_gpu_march_stats(level, out, 4))
        return {"sa_march": out[0], "visc_gf": out[1], "tile_march": out[2]}

    def sync(self):
        self._chk(self.lib.adflow_gpu_sync())
