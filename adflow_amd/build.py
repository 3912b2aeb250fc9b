"""Build libadflow_gpu.so (hand-written HIP for gfx950) in-tree with hipcc.

`python -m adflow_amd.build` or `adflow_amd.build.build_lib()`.  The .so stays
inside the repository (adflow_amd/lib/) so it travels to the GPU box with the
source snapshot; it is git-ignored.
"""
from __future__ import annotations

import concurrent.futures
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libadflow_gpu.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "adflow_gpu.h")]
    jobs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        # kernels_ad.hip #includes the gather-kernel SOURCES a second time (dual numbers): it is stale when any of them is
        deps = [s] + hdrs + (srcs if os.path.basename(s) == "kernels_ad.hip" else [])
        if force or _stale(o, deps):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        tmp = f"{o}.tmp.{os.getpid()}"
        cmd = [HIPCC] + CFLAGS + ["-c", s, "-o", tmp]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0:
            os.replace(tmp, o)
        elif os.path.exists(tmp):
            os.remove(tmp)
        return s, r.returncode, r.stdout + r.stderr

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for s, rc, out in ex.map(cc, jobs):
                if verbose and out.strip():
                    print(out, file=sys.stderr)
                if rc != 0:
                    raise RuntimeError(f"hipcc failed for {s}:\n{out}")
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if jobs or force or _stale(LIB, objs):
        # -Bsymbolic-functions: calls between the library's own entry points (adflow_gpu_mg_cycle -> adflow_gpu_rk_smooth ...) bind
        # inside the library and cannot be interposed by another definition of the same name in the process
        # (linked under another name and renamed: a process that loads the library never sees a half-written file)
        tmp = f"{LIB}.tmp.{os.getpid()}"
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-Wl,-Bsymbolic-functions", "-o", tmp] + objs + ["-L/opt/rocm/lib", "-lrccl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
