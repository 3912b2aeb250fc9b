"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Compile the HIP kernel sources of adflow_amd/csrc with g++ against the host
emulation header in this directory -> tests/hostsim/libadflow_hostsim.so.
Lets the CPU-only CI exercise the kernel logic against the oracle.  adflow_amd
never loads this library."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, "libadflow_hostsim.so")


def build(force=False):
    srcs = sorted(glob.glob(os.path.join(ROOT, "adflow_amd", "csrc", "*.hip"))) + [os.path.join(HERE, "hostsim.cpp")]
    deps = srcs + [os.path.abspath(__file__)] + glob.glob(os.path.join(ROOT, "adflow_amd", "csrc", "*.h")) + \
        [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "adflow_gpu.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    # (linked into a file of its own and renamed: several pytest-xdist workers may find the library stale at the same time, and a
    # reader must never see a half-written one)
    tmp = f"{LIB}.tmp.{os.getpid()}"
    cmd = ["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-Wl,-Bsymbolic-functions", "-I", HERE,
           "-DADFLOW_NO_RCCL", "-Wno-unknown-pragmas", "-o", tmp]
    for s in srcs:
        cmd += ["-x", "c++", s]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("hostsim build failed:\n" + r.stdout + r.stderr)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
