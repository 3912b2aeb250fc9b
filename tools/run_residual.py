#!/usr/bin/env python
"""Profiling driver: N residual evaluations on one block (for rocprofv3 passes).
usage: run_residual.py [euler|rans] nx ny nz nevals [scheme]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adflow_amd.engine import Engine
from adflow_amd.params import FlowParams, RANSEquations
from adflow_amd.synth import make_block

kind = sys.argv[1] if len(sys.argv) > 1 else "euler"
nx, ny, nz = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (128, 128, 128)
nev = int(sys.argv[5]) if len(sys.argv) > 5 else 10
sd = int(sys.argv[6]) if len(sys.argv) > 6 else 1
prm = FlowParams(equations=RANSEquations if kind == "rans" else 1, spaceDiscr=sd)
eng = Engine(0)
eng.set_options(prm)
blk = make_block(nx, ny, nz, prm, seed=1, stretch_k=3.0 if kind == "rans" else 1.0)
eng.register(blk)
for _ in range(nev):
    eng.blocketteRes(1, True, True, kind == "rans")
eng.close()
