#!/usr/bin/env python
"""Event timing of single entry points on the MI355X (8 x 128^3 Euler blocks): time step with and without the
directional scaling, halo exchange, stage update.  usage: time_kernels.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adflow_amd.engine import Engine
from adflow_amd.params import FlowParams
from adflow_amd.synth import make_block

prm = FlowParams()
eng = Engine(0)
eng.set_options(prm)
blk = make_block(128, 128, 128, prm, seed=1)
for nn in range(1, 9):
    eng.register(blk, nn=nn)


def timeit(name, fn, n=20):
    for _ in range(3):
        fn()
    eng.event_record(0)
    for _ in range(n):
        fn()
    eng.event_record(1)
    eng.sync()
    print(f"{name}: {eng.event_elapsed_ms(0, 1) / n:.4f} ms", flush=True)


timeit("timeStep dirScaling=on ", lambda: eng.timeStep(1, False))
timeit("timeStep onlyRadii      ", lambda: eng.timeStep(1, True))
eng.set_options(prm.replace(dirScaling=False))
timeit("timeStep dirScaling=off", lambda: eng.timeStep(1, False))
eng.close()
