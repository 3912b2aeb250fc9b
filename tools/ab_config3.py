#!/usr/bin/env python
"""A/B of tuning knobs on the config-3 iteration (D-ADI x3 + SA DDADI x3, scalar JST RANS) and the scalar-JST RANS evaluation.
usage: ab_config3.py "key=val key=val" "key=val" ...   (one set per argument; "" = defaults)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from adflow_amd.engine import Engine  # noqa: E402
from adflow_amd.params import DADI, noResAveraging  # noqa: E402


class A:
    steps, warmup, workload, min_seconds, tuning = 10, 2, "crm_rans_sa_upwind_8x160x128x64", 0.5, []


def main():
    import torch
    eng = Engine(0)
    job = bench.Job(A, A.workload, eng, 0, 1)
    prm = job.prm.replace(spaceDiscr=1, smoother=DADI, nSubiterations=3, nSubIterTurb=3, cfl=1.5, resAveraging=noResAveraging)

    def barrier():
        torch.cuda.synchronize()
        eng.sync()
    for arg in sys.argv[1:] or [""]:
        kv = dict(x.split("=") for x in arg.split())
        for k, v in kv.items():
            eng.set_tuning(k, int(v))
        eng.set_options(prm)
        eng.timeStep(1, False)
        eng.residual(1, 0)
        for _ in range(2):
            eng.executeMGCycle([0])
        s3, _, _ = bench.timed(eng, lambda: eng.executeMGCycle([0]), 5, barrier, 0.5)
        eng.set_async(True)
        for _ in range(3):
            job.step()
        se, _, _ = bench.timed(eng, job.step, 20, barrier, 0.5)
        eng.set_async(False)
        print(f"[{arg or 'defaults'}] config-3 iteration {s3 * 1e3:.3f} ms; scalar-JST RANS evaluation {se * 1e3:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
