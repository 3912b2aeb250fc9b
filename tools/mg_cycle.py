#!/usr/bin/env python
"""The 3-level W cycle of BASELINE config 1 (Euler JST 8 x 128^3, RK5 + alternate residual averaging) alone, for a kernel trace.
usage: mg_cycle.py [n cycles] [key=val ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from adflow_amd.engine import Engine  # noqa: E402
from adflow_amd.params import RungeKutta, alternateResAveraging  # noqa: E402


class A:
    steps, warmup, workload, min_seconds, tuning = 10, 2, "euler_jst_8x128", 0.5, []


def main():
    import torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    eng = Engine(0)
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        eng.set_tuning(k, int(v))
    j2 = bench.Job(A, "euler_jst_8x128", eng, 0, 1, levels=3)
    eng.set_options(j2.prm.replace(smoother=RungeKutta, resAveraging=alternateResAveraging))
    cyc = bench.w_cycle(3)
    eng.timeStep(1, False)
    eng.residual(1, 0)
    for _ in range(2):
        eng.executeMGCycle(cyc)
    torch.cuda.synchronize(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.executeMGCycle(cyc)
    torch.cuda.synchronize(); eng.sync()
    print(f"3w MG cycle: {(time.perf_counter() - t0) / n * 1e3:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
