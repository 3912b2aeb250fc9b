"""ctypes binding of the C-ABI in include/adflow_gpu.h.

This is the Python twin of the ISO_C_BINDING interface module a Fortran host
uses (adflow_amd/fortran/adflow_gpu_shim.F90); it exists so that the parity
tests and the benchmark drive exactly the entry points the reference's shell
routines would call.  There is NO CPU fallback: if the HIP library is missing
or no GPU is visible every call raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int8, c_int32, c_uint, c_void_p
from typing import Optional

import numpy as np

from . import params as P

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libadflow_gpu.so")
MAX_RK = 8


class AdflowOpts(ctypes.Structure):
    _fields_ = [
        ("equations", c_int32), ("turbModel", c_int32), ("turbProd", c_int32),
        ("useQCR", c_int32), ("useRotationSA", c_int32), ("useft2SA", c_int32),
        ("spaceDiscr", c_int32), ("spaceDiscrCoarse", c_int32), ("limiter", c_int32), ("orderTurb", c_int32),
        ("dirScaling", c_int32),
        ("smoother", c_int32), ("nRKStages", c_int32), ("resAveraging", c_int32), ("nSubiterations", c_int32),
        ("nSubIterTurb", c_int32),
        ("groundLevel", c_int32),
        ("turbRelax", c_int32),
        ("eulerWallBCTreatment", c_int32),
        ("viscWallBCTreatment", c_int32),
        ("outflowTreatment", c_int32),
        ("hScalingInlet", c_int32), ("unsupported", c_int32), ("lowSpeedPreconditioner", c_int32),
        ("exchangePressureEarly", c_int32), ("reserved_i", c_int32),
        ("gammaConstant", c_double), ("prandtl", c_double), ("prandtlTurb", c_double),
        ("SSuthDim", c_double), ("muSuthDim", c_double), ("TSuthDim", c_double),
        ("SAKappa", c_double), ("SAcb1", c_double), ("SAcb2", c_double), ("SAsigma", c_double), ("SAcv1", c_double),
        ("SAcw1", c_double), ("SAcw2", c_double), ("SAcw3", c_double), ("SAct1", c_double), ("SAct2", c_double),
        ("SAct3", c_double), ("SAct4", c_double), ("SAcrot", c_double),
        ("vis2", c_double), ("vis4", c_double), ("vis2Coarse", c_double), ("adis", c_double),
        ("acousticScaleFactor", c_double), ("kappaCoef", c_double),
        ("cfl", c_double), ("cflCoarse", c_double), ("cflLimit", c_double), ("fcoll", c_double), ("smoop", c_double),
        ("alfaTurb", c_double), ("betaTurb", c_double), ("turbResScale", c_double),
        ("etaRK", c_double * MAX_RK), ("cdisRK", c_double * MAX_RK),
        ("gammaInf", c_double), ("pInf", c_double), ("pInfCorr", c_double), ("rhoInf", c_double), ("uInf", c_double),
        ("RGas", c_double), ("muInf", c_double), ("muRef", c_double), ("TRef", c_double), ("timeRef", c_double),
        ("wInf", c_double * 10),
        ("sigma", c_double),
        ("pRef", c_double), ("uRef", c_double), ("LRef", c_double), ("ordersConverged", c_double),
        ("reserved_d", c_double * 3),
    ]


class AdflowBcSubface(ctypes.Structure):
    """adflow_bc_subface (include/adflow_gpu.h): BCType/BCFaceID + the BCData members the flow BCs read"""
    _fields_ = [
        ("bcType", c_int32), ("faceID", c_int32),
        ("icBeg", c_int32), ("icEnd", c_int32), ("jcBeg", c_int32), ("jcEnd", c_int32),
        ("subsonicInletTreatment", c_int32), ("reserved", c_int32),
        ("norm", c_void_p), ("rface", c_void_p), ("uSlip", c_void_p), ("TNS_Wall", c_void_p),
        ("rho", c_void_p), ("velx", c_void_p), ("vely", c_void_p), ("velz", c_void_p), ("ps", c_void_p),
        ("ptInlet", c_void_p), ("ttInlet", c_void_p), ("htInlet", c_void_p),
        ("flowXdirInlet", c_void_p), ("flowYdirInlet", c_void_p), ("flowZdirInlet", c_void_p), ("turbInlet", c_void_p),
        ("symNorm", c_double * 3),
    ]


BC_ARRAYS = ("norm", "rface", "uSlip", "TNS_Wall", "rho", "velx", "vely", "velz", "ps", "ptInlet", "ttInlet", "htInlet",
             "flowXdirInlet", "flowYdirInlet", "flowZdirInlet", "turbInlet")


# BCType / BCFaceID values (src/modules/constants.F90:257-297)
BC_SYMM, BC_NSWALL_ADIABATIC, BC_NSWALL_ISOTHERMAL, BC_EULERWALL, BC_FARFIELD = -1, -3, -4, -5, -6
BC_SUPERSONIC_INFLOW, BC_SUPERSONIC_OUTFLOW, BC_EXTRAP = -7, -9, -15
BC_SYMM_POLAR, BC_SUBSONIC_INFLOW, BC_SUBSONIC_OUTFLOW, BC_MASSBLEED_OUTFLOW = -2, -8, -10, -12
IMIN, IMAX, JMIN, JMAX, KMIN, KMAX = 1, 2, 3, 4, 5, 6


class AdflowBlockDesc(ctypes.Structure):
    _fields_ = [
        ("nx", c_int32), ("ny", c_int32), ("nz", c_int32), ("nw", c_int32), ("rightHanded", c_int32),
        ("reserved", c_int32),
        ("w", c_void_p), ("p", c_void_p), ("gamma", c_void_p), ("rlv", c_void_p), ("rev", c_void_p),
        ("x", c_void_p), ("sI", c_void_p), ("sJ", c_void_p), ("sK", c_void_p),
        ("vol", c_void_p), ("volRef", c_void_p), ("d2Wall", c_void_p),
        ("porI", c_void_p), ("porJ", c_void_p), ("porK", c_void_p), ("iblank", c_void_p),
        ("dw", c_void_p), ("fw", c_void_p), ("dtl", c_void_p), ("radI", c_void_p), ("radJ", c_void_p),
        ("radK", c_void_p),
        ("w1", c_void_p), ("p1", c_void_p), ("wr", c_void_p),
        ("mgIFine", c_void_p), ("mgJFine", c_void_p), ("mgKFine", c_void_p),
        ("mgIWeight", c_void_p), ("mgJWeight", c_void_p), ("mgKWeight", c_void_p),
        ("mgICoarse", c_void_p), ("mgJCoarse", c_void_p), ("mgKCoarse", c_void_p),
        ("sFaceI", c_void_p), ("sFaceJ", c_void_p), ("sFaceK", c_void_p),
        ("rotRate", c_double * 3), ("addGridVelocities", c_int32), ("blockIsMoving", c_int32),
    ]


class AdflowActuatorRegion(ctypes.Structure):
    """adflow_actuator_region: actuatorRegionType of actuatorRegionData.F90"""
    _fields_ = [
        ("nCellIDs", c_int32), ("reserved", c_int32), ("block", c_void_p), ("cellIDs", c_void_p),
        ("force", c_double * 3), ("heat", c_double), ("volume", c_double), ("relaxStart", c_double), ("relaxEnd", c_double),
    ]


class AdflowPeriodicData(ctypes.Structure):
    """adflow_periodic_data: periodicDataType of communication.F90"""
    _fields_ = [
        ("rotMatrix", c_double * 9), ("rotCenter", c_double * 3), ("translation", c_double * 3),
        ("nHalos", c_int32), ("reserved", c_int32), ("block", c_void_p), ("indices", c_void_p),
    ]


class AdflowCommPattern(ctypes.Structure):
    _fields_ = [
        ("ncopy", c_int32),
        ("donorBlock", c_void_p), ("donorIndices", c_void_p), ("haloBlock", c_void_p), ("haloIndices", c_void_p),
        ("nProcSend", c_int32), ("sendProc", c_void_p), ("nsendCum", c_void_p), ("sendBlock", c_void_p),
        ("sendIndices", c_void_p),
        ("nProcRecv", c_int32), ("recvProc", c_void_p), ("nrecvCum", c_void_p), ("recvBlock", c_void_p),
        ("recvIndices", c_void_p),
    ]


BC_CALLBACK = ctypes.CFUNCTYPE(None, c_int, c_int)

# array identifiers (include/adflow_gpu.h)
(ARR_W, ARR_P, ARR_GAMMA, ARR_RLV, ARR_REV, ARR_DW, ARR_FW, ARR_DTL, ARR_RADI, ARR_RADJ, ARR_RADK, ARR_AA,
 ARR_NODAL_GRADS, ARR_WN, ARR_PN, ARR_W1, ARR_P1, ARR_WR, ARR_VOL, ARR_SI, ARR_SJ, ARR_SK, ARR_X, ARR_D2WALL) = range(1, 25)

JAC_PC, JAC_FROZEN_TURB, JAC_TURB_ONLY, JAC_VISC_PC, JAC_USE_AD = 1, 2, 4, 8, 16     # include/adflow_gpu.h
RES_UPDATE_INTERMED, RES_FLOW, RES_TURB, RES_CLOSURES, RES_HALO = 1, 2, 4, 8, 16

EXPORTS = [
    "adflow_gpu_init", "adflow_gpu_finalize", "adflow_gpu_last_error", "adflow_gpu_device_name",
    "adflow_gpu_comm_unique_id", "adflow_gpu_comm_init",
    "adflow_gpu_block_register", "adflow_gpu_block_release", "adflow_gpu_release_all", "adflow_gpu_upload_geometry", "adflow_gpu_upload_state",
    "adflow_gpu_download_state", "adflow_gpu_download_residual", "adflow_gpu_download_array",
    "adflow_gpu_upload_array", "adflow_gpu_set_options",
    "adflow_gpu_time_step", "adflow_gpu_initres", "adflow_gpu_residual", "adflow_gpu_block_res",
    "adflow_gpu_rk_smooth", "adflow_gpu_dadi_smooth", "adflow_gpu_halo_exchange", "adflow_gpu_res_norms",
    "adflow_gpu_sa_solve", "adflow_gpu_set_turb_bc_callback",
    "adflow_gpu_set_w_vec", "adflow_gpu_get_r_vec", "adflow_gpu_get_res", "adflow_gpu_nk_residual",
    "adflow_gpu_nk_residual_dev",
    "adflow_gpu_transfer_to_coarse", "adflow_gpu_transfer_to_fine", "adflow_gpu_mg_cycle",
    "adflow_gpu_comm_register", "adflow_gpu_comm_info", "adflow_gpu_halo_slot_info", "adflow_gpu_halo_pack", "adflow_gpu_halo_unpack",
    "adflow_gpu_halo_local_copy", "adflow_gpu_set_bc_callback", "adflow_gpu_bc_register", "adflow_gpu_apply_all_bc", "adflow_gpu_download_wall_stress", "adflow_gpu_abi_sizes2", "adflow_gpu_xhalo", "adflow_gpu_actuator_register", "adflow_gpu_comm_register_periodic", "adflow_gpu_coarse_coordinates", "adflow_gpu_exchange_coor",
    "adflow_gpu_upload_coordinates", "adflow_gpu_update_geometry", "adflow_gpu_reference_shock_sensor",
    "adflow_gpu_wall_distance_register", "adflow_gpu_update_wall_distances",
    "adflow_gpu_fd_jacobian", "adflow_gpu_release_workspace", "adflow_gpu_selftest_math", "adflow_gpu_jacobian_info", "adflow_gpu_download_jacobian", "adflow_gpu_download_jacobian_rows",
    "adflow_gpu_event_record", "adflow_gpu_event_elapsed_ms", "adflow_gpu_sync", "adflow_gpu_set_async",
    "adflow_gpu_abi_sizes", "adflow_gpu_set_tuning", "adflow_gpu_march_stats",
]

_libs = {}


class AdflowGpuError(RuntimeError):
    pass


def load(path: Optional[str] = None) -> ctypes.CDLL:
    """dlopen the HIP library; raises (never falls back) if it is absent.
    `path` is only ever passed by the test-suite (kernel-logic emulator)."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise AdflowGpuError(f"{path} not built: run `python -m adflow_amd.build` (no CPU fallback exists)")
    lib = ctypes.CDLL(path)
    lib.adflow_gpu_last_error.restype = c_char_p
    lib.adflow_gpu_init.argtypes = [c_int]
    lib.adflow_gpu_device_name.argtypes = [c_char_p, c_int]
    lib.adflow_gpu_block_register.argtypes = [c_int, c_int, c_int, POINTER(AdflowBlockDesc)]
    for n in ("adflow_gpu_block_release", "adflow_gpu_upload_geometry", "adflow_gpu_upload_state", "adflow_gpu_download_state",
              "adflow_gpu_download_residual"):
        getattr(lib, n).argtypes = [c_int, c_int, c_int]
    lib.adflow_gpu_download_array.argtypes = [c_int, c_int, c_int, c_int, c_void_p]
    lib.adflow_gpu_upload_array.argtypes = [c_int, c_int, c_int, c_int, c_void_p]
    lib.adflow_gpu_set_options.argtypes = [POINTER(AdflowOpts)]
    lib.adflow_gpu_time_step.argtypes = [c_int, c_int]
    lib.adflow_gpu_initres.argtypes = [c_int, c_int, c_int]
    lib.adflow_gpu_residual.argtypes = [c_int, c_int]
    lib.adflow_gpu_block_res.argtypes = [c_int, c_uint]
    lib.adflow_gpu_set_async.argtypes = [c_int]
    lib.adflow_gpu_bc_register.argtypes = [c_int, c_int, c_int, c_int, c_int, POINTER(AdflowBcSubface)]
    lib.adflow_gpu_apply_all_bc.argtypes = [c_int, c_int]
    lib.adflow_gpu_download_wall_stress.argtypes = [c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.adflow_gpu_upload_coordinates.argtypes = [c_int, c_int, c_int]
    lib.adflow_gpu_update_geometry.argtypes = [c_int]
    lib.adflow_gpu_xhalo.argtypes = [c_int]
    lib.adflow_gpu_actuator_register.argtypes = [c_int, POINTER(AdflowActuatorRegion)]
    lib.adflow_gpu_comm_register_periodic.argtypes = [c_int, c_int, c_int, POINTER(AdflowPeriodicData)]
    lib.adflow_gpu_coarse_coordinates.argtypes = [c_int]
    lib.adflow_gpu_exchange_coor.argtypes = [c_int]
    lib.adflow_gpu_reference_shock_sensor.argtypes = [c_int]
    lib.adflow_gpu_wall_distance_register.argtypes = [c_int, c_int, c_int, c_void_p, c_void_p]
    lib.adflow_gpu_update_wall_distances.argtypes = [c_int, c_void_p, ctypes.c_int64]
    lib.adflow_gpu_fd_jacobian.argtypes = [c_int, c_uint, c_double]
    lib.adflow_gpu_release_workspace.argtypes = [ctypes.POINTER(ctypes.c_int64)]
    lib.adflow_gpu_selftest_math.argtypes = [c_int, c_void_p, c_void_p, ctypes.c_int64, c_void_p, c_void_p]
    lib.adflow_gpu_jacobian_info.argtypes = [POINTER(ctypes.c_int32), POINTER(ctypes.c_int32), POINTER(ctypes.c_int32)]
    lib.adflow_gpu_download_jacobian.argtypes = [c_int, c_int, c_int, c_void_p]
    lib.adflow_gpu_download_jacobian_rows.argtypes = [c_int, c_int, c_int, c_void_p]
    lib.adflow_gpu_set_tuning.argtypes = [c_char_p, c_int]
    lib.adflow_gpu_abi_sizes.argtypes = [POINTER(c_int), POINTER(c_int)]
    lib.adflow_gpu_rk_smooth.argtypes = [c_int]
    lib.adflow_gpu_dadi_smooth.argtypes = [c_int]
    lib.adflow_gpu_halo_exchange.argtypes = [c_int] * 6
    lib.adflow_gpu_res_norms.argtypes = [c_int, c_void_p, c_int]
    lib.adflow_gpu_sa_solve.argtypes = [c_int]
    lib.adflow_gpu_set_turb_bc_callback.argtypes = [c_void_p]
    lib.adflow_gpu_set_w_vec.argtypes = [c_void_p, ctypes.c_long]
    lib.adflow_gpu_get_r_vec.argtypes = [c_void_p, ctypes.c_long, c_void_p]
    lib.adflow_gpu_get_res.argtypes = [c_void_p, ctypes.c_long]
    lib.adflow_gpu_nk_residual.argtypes = [c_void_p, c_void_p, ctypes.c_long]
    lib.adflow_gpu_nk_residual_dev.argtypes = [c_void_p, c_void_p, ctypes.c_long]
    lib.adflow_gpu_transfer_to_coarse.argtypes = [c_int]
    lib.adflow_gpu_transfer_to_fine.argtypes = [c_int]
    lib.adflow_gpu_mg_cycle.argtypes = [c_void_p, c_int]
    lib.adflow_gpu_comm_register.argtypes = [c_int, c_int, POINTER(AdflowCommPattern)]
    lib.adflow_gpu_halo_slot_info.argtypes = [c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]
    lib.adflow_gpu_comm_info.argtypes = [POINTER(c_int)] * 4
    lib.adflow_gpu_halo_pack.argtypes = [c_int] * 7 + [c_void_p]
    lib.adflow_gpu_halo_unpack.argtypes = [c_int] * 7 + [c_void_p]
    lib.adflow_gpu_halo_local_copy.argtypes = [c_int] * 6
    lib.adflow_gpu_set_bc_callback.argtypes = [c_void_p]
    lib.adflow_gpu_comm_unique_id.argtypes = [c_void_p]
    lib.adflow_gpu_comm_init.argtypes = [c_int, c_int, c_void_p]
    lib.adflow_gpu_event_record.argtypes = [c_int]
    lib.adflow_gpu_event_elapsed_ms.argtypes = [c_int, c_int, POINTER(c_double)]
    lib.adflow_gpu_march_stats.argtypes = [c_int, POINTER(c_double), c_int]
    so, sd = c_int(), c_int()
    lib.adflow_gpu_abi_sizes(ctypes.byref(so), ctypes.byref(sd))
    if so.value != ctypes.sizeof(AdflowOpts) or sd.value != ctypes.sizeof(AdflowBlockDesc):
        raise AdflowGpuError("ctypes mirror of adflow_opts/adflow_block_desc is out of date with include/adflow_gpu.h")
    _libs[path] = lib
    return lib


def check(rc: int, lib: Optional[ctypes.CDLL] = None) -> None:
    if rc != 0:
        raise AdflowGpuError((lib or load()).adflow_gpu_last_error().decode())


def opts_from_params(prm: P.FlowParams) -> AdflowOpts:
    o = AdflowOpts()
    for name, _ in AdflowOpts._fields_:
        if name.startswith("reserved") or name in ("etaRK", "cdisRK", "wInf"):
            continue
        setattr(o, name, type(getattr(o, name))(getattr(prm, name)))
    for i, v in enumerate(prm.etaRK):
        o.etaRK[i] = v
    for i, v in enumerate(prm.cdisRK):
        o.cdisRK[i] = v
    for i, v in enumerate(prm.wInf()):
        o.wInf[i] = v
    return o


def _ptr(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.flags["F_CONTIGUOUS"], "host arrays must be Fortran-ordered"
    return a.ctypes.data


def desc_from_block(blk) -> AdflowBlockDesc:
    d = AdflowBlockDesc()
    d.nx, d.ny, d.nz, d.nw, d.rightHanded = blk.nx, blk.ny, blk.nz, blk.nw, int(getattr(blk, "rightHanded", True))
    for name in ("w", "p", "gamma", "rlv", "rev", "x", "sI", "sJ", "sK", "vol", "volRef", "d2Wall",
                 "porI", "porJ", "porK", "iblank", "dw", "fw", "dtl", "radI", "radJ", "radK", "w1", "p1", "wr",
                 "mgIFine", "mgJFine", "mgKFine", "mgIWeight", "mgJWeight", "mgKWeight", "mgICoarse", "mgJCoarse",
                 "mgKCoarse", "sFaceI", "sFaceJ", "sFaceK"):
        setattr(d, name, _ptr(blk.a.get(name)))
    d.addGridVelocities = int(blk.a.get("sFaceI") is not None)
    rot = getattr(blk, "rotRate", None)
    d.blockIsMoving = int(rot is not None)
    for m in range(3):
        d.rotRate[m] = float(rot[m]) if rot is not None else 0.0
    return d


def comm_pattern_struct(cp) -> AdflowCommPattern:
    """adflow_amd.topology.CommPattern -> C struct (arrays must outlive the call)."""
    c = AdflowCommPattern()
    c.ncopy = cp.ncopy
    for n in ("donorBlock", "donorIndices", "haloBlock", "haloIndices", "sendProc", "nsendCum", "sendBlock",
              "sendIndices", "recvProc", "nrecvCum", "recvBlock", "recvIndices"):
        a = getattr(cp, n)
        assert a.dtype == np.int32 and (a.ndim == 1 or a.flags["F_CONTIGUOUS"]), n
        setattr(c, n, a.ctypes.data if a.size else None)
    c.nProcSend = int(cp.sendProc.size)
    c.nProcRecv = int(cp.recvProc.size)
    return c
