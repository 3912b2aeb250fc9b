"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.
ctypes front-end of oracle/adflow_oracle.c (plain-C restatement of the Euler
residual path).  build() compiles it with gcc into oracle/_build/."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "adflow_oracle.c")
LIB = os.path.join(HERE, "_build", "libadflow_oracle.so")


class OracleBlock(ctypes.Structure):
    _fields_ = [("nx", ctypes.c_int), ("ny", ctypes.c_int), ("nz", ctypes.c_int), ("nw", ctypes.c_int),
                ("vis2", ctypes.c_double), ("vis4", ctypes.c_double), ("adis", ctypes.c_double),
                ("acousticScaleFactor", ctypes.c_double), ("rFil", ctypes.c_double),
                ("gammaInf", ctypes.c_double), ("pInfCorr", ctypes.c_double), ("rhoInf", ctypes.c_double),
                ("dirScaling", ctypes.c_int), ("onlyRadii", ctypes.c_int), ("iblankUsed", ctypes.c_int)] + \
               [(n, ctypes.c_void_p) for n in ("w", "p", "gamma", "sI", "sJ", "sK", "porI", "porJ", "porK", "iblank",
                                               "radI", "radJ", "radK", "dtl", "dw", "fw")]


def build(force=False) -> str:
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", LIB, SRC, "-lm"], check=True)
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def block_res_euler_scalar(blk, prm):
    """In place on blk: fills dw, fw, radI/J/K, dtl (creates them if missing)."""
    a = blk.a
    ib, jb, kb, ie, je, ke = blk.ib, blk.jb, blk.kb, blk.ie, blk.je, blk.ke
    for name, shape in (("dw", (ib + 1, jb + 1, kb + 1, blk.nw)), ("fw", (ib + 1, jb + 1, kb + 1, 5)),
                        ("dtl", (ie, je, ke)), ("radI", (ie, je, ke)), ("radJ", (ie, je, ke)), ("radK", (ie, je, ke))):
        if name not in a:
            a[name] = np.zeros(shape, order="F")
    o = OracleBlock()
    o.nx, o.ny, o.nz, o.nw = blk.nx, blk.ny, blk.nz, blk.nw
    o.vis2, o.vis4, o.adis, o.acousticScaleFactor, o.rFil = prm.vis2, prm.vis4, prm.adis, prm.acousticScaleFactor, 1.0
    o.gammaInf, o.pInfCorr, o.rhoInf = prm.gammaInf, prm.pInfCorr, prm.rhoInf
    o.dirScaling, o.onlyRadii, o.iblankUsed = int(prm.dirScaling), 0, 1
    for n in ("w", "p", "gamma", "sI", "sJ", "sK", "porI", "porJ", "porK", "iblank", "radI", "radJ", "radK", "dtl",
              "dw", "fw"):
        assert a[n].flags["F_CONTIGUOUS"]
        setattr(o, n, a[n].ctypes.data)
    load().oracle_block_res_euler_scalar(ctypes.byref(o))
    return blk
