#!/usr/bin/env python
"""The device part of the `useSpatial` branch of blocketteRes (volume_block + metric_block + boundaryNormals, then the
face vectors / node sums the next evaluation re-forms) on the north-star blocks, alone.  usage: geom_update.py [n]"""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    eng = Engine(0)
    job = bench.Job(A, A.workload, eng, 0, 1)
    eng.set_options(job.prm)
    job.step()
    eng.update_geometry(1)
    job.step()
    torch.cuda.synchronize(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.update_geometry(1)
    torch.cuda.synchronize(); eng.sync()
    t1 = time.perf_counter()
    for _ in range(n):
        eng.update_geometry(1)
        job.step()
    torch.cuda.synchronize(); eng.sync()
    t2 = time.perf_counter()
    for _ in range(n):
        job.step()
    torch.cuda.synchronize(); eng.sync()
    t3 = time.perf_counter()
    print(f"update_geometry: {(t1 - t0) / n * 1e3:.3f} ms; update + evaluation {(t2 - t1) / n * 1e3:.3f} ms; evaluation {(t3 - t2) / n * 1e3:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
