"""Randomised parity sweep: block sizes, options and entry points drawn at random, every case checked against the reference's
own Fortran (oracle/_ref) at the 1e-10 bar of tests/util.py.  Test infrastructure, not a test module: run it by hand,

    python tests/fuzz_parity.py --cases 200 --seed 7             # kernel-logic emulator (CPU only)
    python tests/fuzz_parity.py --cases 200 --seed 7 --gpu       # the HIP library on cuda:0

It prints one line per case and stops at the first failure with the arguments that reproduce it.  The sizes are chosen around the
tile edges of the marching kernels (60 / 62 produced columns of 64 lanes, 4 rows, k chunks of 32 / 22 planes)."""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import checks  # noqa: E402
from adflow_amd.params import (FlowParams, EulerEquations, NSEquations, RANSEquations, dissScalar, dissMatrix, upwind,  # noqa: E402
                               noLimiter, vanAlbeda, minmod, firstOrder, secondOrder)

EDGE_NX = [1, 2, 3, 5, 58, 59, 60, 61, 62, 63, 64, 65, 119, 120, 121, 124, 125]
EDGE_NY = [1, 2, 3, 4, 5, 7, 8, 9, 12, 13]
EDGE_NZ = [1, 2, 3, 5, 20, 21, 22, 23, 31, 32, 33, 34, 43, 44, 45]


def draw_case(rng):
    eq = int(rng.choice([EulerEquations, NSEquations, RANSEquations], p=[0.25, 0.25, 0.5]))
    sd = int(rng.choice([dissScalar, dissMatrix, upwind], p=[0.3, 0.2, 0.5]))
    big = rng.random() < 0.35
    nx = int(rng.choice(EDGE_NX)) if big else int(rng.integers(1, 14))
    ny = int(rng.choice(EDGE_NY))
    nz = int(rng.choice(EDGE_NZ)) if (big and nx < 70) else int(rng.integers(1, 12))
    kw = dict(equations=eq, spaceDiscr=sd)
    if sd == upwind:
        kw["limiter"] = int(rng.choice([noLimiter, vanAlbeda, minmod]))
        kw["kappaCoef"] = float(rng.choice([1.0 / 3.0, -1.0, 0.0, 0.5]))
    else:
        kw["vis2"] = float(rng.choice([0.25, 0.0, 0.5]))
        kw["vis4"] = float(rng.choice([0.0156, 0.1, 0.03]))
        kw["dirScaling"] = bool(rng.random() < 0.8)
        kw["adis"] = float(rng.choice([0.67, 1.0, 0.5]))
    if eq == RANSEquations:
        kw["useQCR"] = bool(rng.random() < 0.3)
        kw["orderTurb"] = int(rng.choice([firstOrder, secondOrder]))
        kw["useRotationSA"] = bool(rng.random() < 0.2)
        kw["useft2SA"] = bool(rng.random() < 0.7)
    if eq != EulerEquations and rng.random() < 0.4:
        kw["muSuthDim"] = 1.0          # viscous-dominated
    mk = dict(seed=int(rng.integers(1, 10 ** 6)))
    if rng.random() < 0.5:
        mk["stretch_k"] = float(rng.choice([1.5, 2.0, 3.0]))
    if rng.random() < 0.3:
        mk["holes"] = 0.05
    if rng.random() < 0.15:
        mk["left_handed"] = True
    entry = str(rng.choice(["block_res", "blockette", "blockette_intermed"], p=[0.6, 0.25, 0.15]))
    return (nx, ny, nz), kw, mk, entry


def run_case(engine, dims, kw, mk, entry):
    prm = FlowParams(**kw)
    mk = dict(mk)
    seed = mk.pop("seed")
    if entry == "block_res" or not kw.get("dirScaling", True):
        # (blocketteResCore scales the dissipation unconditionally: the dirScaling = F case has blockResCore as its reference)
        checks.check_block_res(engine, dims, prm, seed=seed, **mk)
    else:
        checks.check_block_res_vs_blockette(engine, dims, prm, update_intermed=(entry == "blockette_intermed"), seed=seed, **mk)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--gpu", action="store_true", help="the HIP library on cuda:0 instead of the emulator")
    ap.add_argument("--only", type=int, default=-1, help="run this case index only")
    a = ap.parse_args()
    from adflow_amd.engine import Engine
    if a.gpu:
        eng = Engine(0)
    else:
        from hostsim.build import build
        eng = Engine(0, _lib_path=build())
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    nfail = 0
    for n in range(a.cases):
        dims, kw, mk, entry = draw_case(rng)
        if a.only >= 0 and n != a.only:
            continue
        try:
            run_case(eng, dims, kw, mk, entry)
            print(f"[{n:4d}] ok   {dims} {entry} {kw} {mk}", flush=True)
        except AssertionError as ex:
            nfail += 1
            print(f"[{n:4d}] FAIL {dims} {entry} {kw} {mk}: {ex}", flush=True)
            print(f"reproduce: python tests/fuzz_parity.py --seed {a.seed} --cases {a.cases} --only {n}" + (" --gpu" if a.gpu else ""))
            break
    print(f"{a.cases if not nfail else n + 1} cases, {nfail} failures, {time.time() - t0:.0f} s")
    eng.close()
    return 1 if nfail else 0


if __name__ == "__main__":
    sys.exit(main())
