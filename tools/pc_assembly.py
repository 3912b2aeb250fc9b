#!/usr/bin/env python
"""The preconditioner matrix of NK / ANK on the north-star blocks, alone (for a kernel trace): setupStateResidualMatrix(usePC = T).
usage: pc_assembly.py [n]   (n timed assemblies, default 2)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from adflow_amd.engine import Engine  # noqa: E402


class A:
    steps, warmup, workload, min_seconds, tuning = 10, 2, "crm_rans_sa_upwind_8x160x128x64", 0.5, []


def main():
    import torch
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    eng = Engine(0)
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        eng.set_tuning(k, int(v))
    job = bench.Job(A, A.workload, eng, 0, 1)
    eng.set_options(job.prm)
    job.step()
    eng.setupStateResidualMatrix(1, usePC=True)
    torch.cuda.synchronize(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.setupStateResidualMatrix(1, usePC=True)
    torch.cuda.synchronize(); eng.sync()
    print(f"PC matrix assembly: {(time.perf_counter() - t0) / n * 1e3:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
