"""GPU parity: Euler residual (central + scalar JST) and time step vs the
reference's own Fortran (oracle/_ref), through the C-ABI."""
import numpy as np
import pytest

from adflow_amd import capi
from adflow_amd.params import FlowParams
from adflow_amd.synth import make_block
from util import TOL, rel_err, owned, ref_block_res

pytestmark = pytest.mark.gpu

# each test uses its own multigrid level number as a namespace so blocks of
# different tests never mix (an entry point acts on all blocks of a level)
CASES = [
    # (level-namespace, nx, ny, nz)   BASELINE configs 1-2 parity sizes + ragged sizes
    (11, 96, 32, 2),
    (12, 16, 14, 9),
    (13, 5, 3, 1),
    (14, 67, 6, 5),
    (15, 1, 1, 1),
]


@pytest.mark.parametrize("lvl,nx,ny,nz", CASES)
def test_euler_scalar_block_res(engine, lvl, nx, ny, nz):
    from oracle import ref
    prm = FlowParams(currentLevel=lvl, groundLevel=lvl)
    blk = make_block(nx, ny, nz, prm, seed=1000 + lvl)
    r = ref_block_res(blk, prm.replace(currentLevel=1, groundLevel=1), True, True, False)
    engine.set_options(prm)
    engine.register(blk, nn=1, level=lvl)
    engine.blocketteRes(level=lvl, updateIntermed=True, flowRes=True, turbRes=False)
    dw = engine.download_residual(1, lvl)
    for l in range(5):
        e = rel_err(owned(blk, dw[..., l]), owned(r, r["dw"][..., l]))
        assert e <= TOL, (l, e)
    for which, name in ((capi.ARR_RADI, "radI"), (capi.ARR_RADJ, "radJ"), (capi.ARR_RADK, "radK"),
                        (capi.ARR_DTL, "dtl")):
        out = np.zeros_like(r[name])
        engine.download_array(which, out, 1, lvl)
        if name == "dtl":
            e = rel_err(out[1:-1, 1:-1, 1:-1], r[name][1:-1, 1:-1, 1:-1])
        else:
            e = rel_err(out, r[name])
        assert e <= TOL, (name, e)
