#!/usr/bin/env python
"""applyAllBC_block on the device at north-star block size: 8 blocks 160x128x64, every block with six physical faces (adiabatic
wall at kMin, symmetry at jMin / jMax, farfield elsewhere), alone; then one evaluation with the boundary conditions inside
(blocketteRes default flags).  usage: bc_apply.py [n]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adflow_amd.engine import Engine  # noqa: E402
from adflow_amd.params import FlowParams, RANSEquations, upwind, vanAlbeda  # noqa: E402
from adflow_amd.synth import make_block, make_bocos  # noqa: E402


def main():
    import torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    eng = Engine(0)
    prm = FlowParams(equations=RANSEquations, spaceDiscr=upwind, limiter=vanAlbeda)
    eng.set_options(prm)
    spec = {1: -6, 2: -6, 3: -1, 4: -1, 5: -3, 6: -6}
    for nn in range(1, 9):
        blk = make_block(160, 128, 64, prm, seed=100 + nn, stretch_k=3.0)
        faces, nvisc = make_bocos(blk, prm, spec, seed=200 + nn)
        eng.register(blk, nn=nn, level=1)
        eng.bc_register(faces, nvisc, nn=nn, level=1)
        print(f"block {nn} registered", flush=True)
    eng.applyAllBC(1, True)
    eng.blocketteRes(1, False, True, True)
    torch.cuda.synchronize(); eng.sync()
    eng.set_async(True)
    t0 = time.perf_counter()
    for _ in range(n):
        eng.applyAllBC(1, True)
    torch.cuda.synchronize(); eng.sync()
    t1 = time.perf_counter()
    for _ in range(n):
        eng.blocketteRes(1, False, True, True)
    torch.cuda.synchronize(); eng.sync()
    t2 = time.perf_counter()
    eng.set_async(False)
    print(f"applyAllBC: {(t1 - t0) / n * 1e3:.3f} ms; blocketteRes (closures + BCs + core): {(t2 - t1) / n * 1e3:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
