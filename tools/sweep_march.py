#!/usr/bin/env python
"""A/B timing of the k-marching Euler residual kernel variants on the MI355X.
usage: sweep_march.py [nblocks nx ny nz]   (default 8 128 128 128)
Each configuration (adflow_gpu_set_tuning knobs) is timed with HIP events on the
engine's stream over 20 residual() calls after 3 warm-up calls."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adflow_amd.engine import Engine
from adflow_amd.params import FlowParams, DADI, noResAveraging
from adflow_amd.synth import make_block

only = None
if "--only" in sys.argv:          # --only pipe,kch,stage : a single configuration (for rocprofv3 --pmc passes)
    ix = sys.argv.index("--only")
    only = [int(x) for x in sys.argv[ix + 1].split(",")]
    del sys.argv[ix:ix + 2]
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nx, ny, nz = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (128, 128, 128)
prm = FlowParams()
eng = Engine(0)
eng.set_options(prm)
blk = make_block(nx, ny, nz, prm, seed=1)
for nn in range(1, nblk + 1):
    eng.register(blk, nn=nn)
eng.timeStep(1, False)
ncell = nblk * nx * ny * nz
CONFIGS = [
    dict(march_pipe=2, march_kch=32, march_by=4),
    dict(march_pipe=2, march_kch=32, march_by=8),
    dict(march_pipe=2, march_kch=64, march_by=8),
    dict(march_pipe=2, march_kch=32, march_by=4),
]
if only:
    CONFIGS = [dict(march_pipe=only[0], march_kch=only[1])]
for stage in ((only[2],) if only else (0, 1)):          # 0: rFil = 1, no persistent fw (D-ADI / blockette form), 1: RK stage with persistent fw
    eng.set_options(FlowParams(smoother=DADI, resAveraging=noResAveraging) if stage == 0 else prm)
    for cfg in CONFIGS:
        for k, v in cfg.items():
            eng.set_tuning(k, v)
        for _ in range(3):
            eng.residual(1, 0 if stage == 0 else 1)
        eng.event_record(0)
        n = 5 if only else 20
        for _ in range(n):
            eng.residual(1, 0 if stage == 0 else 1)
        eng.event_record(1)
        eng.sync()
        ms = eng.event_elapsed_ms(0, 1) / n
        print(f"stage={stage} {cfg} ms/eval={ms:.4f} Gcells/s={ncell / ms * 1e-6:.3f} "
              f"alg GB/s={ncell * 175 / ms * 1e-6:.0f}", flush=True)
eng.close()
