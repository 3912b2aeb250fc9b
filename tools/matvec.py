#!/usr/bin/env python
"""The matrix-free matvec of BASELINE config 4 (FormFunction_mf: setW + blocketteRes + setRVec, device vectors) alone, for a trace.
usage: matvec.py [n] [key=val ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from adflow_amd import capi  # noqa: E402
from adflow_amd.engine import Engine  # noqa: E402


class A:
    steps, warmup, workload, min_seconds, tuning = 10, 2, "crm_rans_sa_upwind_8x160x128x64", 0.5, []


def main():
    import torch
    n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    eng = Engine(0)
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        eng.set_tuning(k, int(v))
    job = bench.Job(A, A.workload, eng, 0, 1, keep_w=True)
    eng.set_options(job.prm)
    job.step()
    n = sum(v.size for v in job.wvec)
    w0 = torch.from_numpy(np.concatenate(job.wvec)).cuda()
    gen = torch.Generator(device="cuda").manual_seed(5)
    vk = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) - 0.5
    vk /= vk.norm()
    wk = w0 + 1e-7 * vk
    rv = torch.empty_like(w0)
    eng.set_async(True)
    for _ in range(3):
        capi.check(eng.lib.adflow_gpu_nk_residual_dev(wk.data_ptr(), rv.data_ptr(), n), eng.lib)
    torch.cuda.synchronize(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(n_it):
        capi.check(eng.lib.adflow_gpu_nk_residual_dev(wk.data_ptr(), rv.data_ptr(), n), eng.lib)
    torch.cuda.synchronize(); eng.sync()
    print(f"matvec: {(time.perf_counter() - t0) / n_it * 1e3:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
