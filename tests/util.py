"""Shared helpers of the parity tests (test infrastructure)."""
import numpy as np

from adflow_amd import capi
from adflow_amd.params import FlowParams
from adflow_amd.synth import make_block

TOL = 1e-10   # BASELINE.json north_star: residual matches the reference to <= 1e-10 relative


def rel_err(a, b):
    """max|a-b| / max|b| (the parity measure of BASELINE.md §3)."""
    den = np.abs(b).max()
    if den == 0.0:
        return np.abs(a).max()
    return np.abs(a - b).max() / den


def owned(blk, arr):
    return arr[2:blk.il + 1, 2:blk.jl + 1, 2:blk.kl + 1]


def ref_block_res(blk, prm, update_intermed=True, flow=True, turb=True):
    """Reference residual (blockette::blockResCore sequence) on a COPY of blk;
    returns the copy holding dw, fw, dtl, radI/J/K, ..."""
    from oracle import ref
    b = blk.copy()
    ref.bind_block(b, prm)
    ref.block_res_core(update_intermed, flow, turb)
    return b


_next_nn = [100]


def gpu_register(engine, blk, prm, level=1):
    nn = _next_nn[0]
    _next_nn[0] += 1
    engine.set_options(prm)
    engine.register(blk, nn=nn, level=level)
    return nn
