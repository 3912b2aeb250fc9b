#!/usr/bin/env python
"""Main-loop instruction counts of the marching kernels of one RANS-SA evaluation -> profiles/isa_counts.json.

Runs on the build machine (hipcc cross-compiles gfx950 without a GPU): compiles the kernel sources to assembly, finds the
march loop of every kernel (tools/isa_count.py) and records VALU / FP64 / transcendental / memory instruction counts per
wavefront and plane, with the git hash of the sources.  bench.py multiplies them with the wavefront-steps of the workload
(adflow_gpu_march_stats) to price the evaluation against the FP64 VALU issue rate (`roofline.fp64_issue`).
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from isa_count import main_loop  # noqa: E402

KERNELS = {
    # key of adflow_gpu_march_stats : (source, regex of the mangled name, what)
    "sa_march": ("kernels_sa_march.hip", r"k_sa_marchILb0ELb0ELb0EEv", "k_sa_march<false>: Spalart-Allmaras residual"),
    "visc_gf": ("kernels_viscous.hip", r"k_visc_gfILb0ELb1ELb0EEv", "k_visc_gf<false,true,false>: nodal gradients + viscous fluxes"),
    "visc_gf_qcr": ("kernels_viscous.hip", r"k_visc_gfILb1ELb1ELb0EEv", "k_visc_gf<true,true,false>: the same with QCR"),
    "roe_march": ("kernels_roe_march.hip", r"k_roe_marchILi3ELb0ELb1ELb1ELb0EEv", "k_roe_march<vanAlbada,.,FINAL,ADDV>: central + Roe upwind"),
    "matrix_march": ("kernels_inviscid_march.hip", r"k_inviscid_marchILi2E", "k_inviscid_march<matrix,...> (first instantiation found)"),
    "euler_march": ("kernels_euler_march.hip", r"k_euler_march_p", "k_euler_march_p (first instantiation found)"),
    "pc_march": ("kernels_pc_march.hip", r"k_pc_marchILb1EEv", "k_pc_march<SNAP>: first-order Roe + thin-layer viscous flux of the preconditioner matrix (static count: the fifth-face block a wave executes in one plane of four included)"),
}


def main():
    out = {"git": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip(),
           "note": "per wavefront and march step (one k plane); issue_cycles = 4 x VALU + 12 more per f64 transcendental seed "
                   "(quarter rate); a CDNA SIMD issues one VALU instruction of a 64-wide wavefront every 4 cycles",
           "kernels": {}}
    srcs = sorted({v[0] for v in KERNELS.values()})
    with tempfile.TemporaryDirectory() as td:
        asm = {}
        for s in srcs:
            o = os.path.join(td, s + ".s")
            # RM_COUNT_NO_CLAMP: the Roe march without the block a lane enters only where a difference lies inside the limiter's
            # epsLim clamp (uniform flow); register / scratch figures of the shipped kernel are taken from the library's own build
            r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                                "-DRM_COUNT_NO_CLAMP", os.path.join(ROOT, "adflow_amd", "csrc", s), "-o", o], capture_output=True, text=True)
            if r.returncode != 0:
                raise SystemExit(r.stderr[-2000:])
            asm[s] = o
        for key, (src, pat, what) in KERNELS.items():
            r = main_loop(asm[src], pat)
            if r is None:
                continue
            r["what"] = what
            r["kernel"] = r["kernel"].split(":")[0]
            out["kernels"][key] = r
        # k_roe_march: the block of the fifth j face is executed by a wave in ONE plane of four (the role rotates over the four waves):
        # the loop is compiled again without it and the average step priced as  without + (with - without) / 4
        o = os.path.join(td, "roe_nofifth.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                            "-DRM_COUNT_NO_FIFTH", "-DRM_COUNT_NO_CLAMP", os.path.join(ROOT, "adflow_amd", "csrc", "kernels_roe_march.hip"), "-o", o],
                           capture_output=True, text=True)
        if r.returncode == 0 and "roe_march" in out["kernels"]:
            base = main_loop(o, KERNELS["roe_march"][1])
            full = out["kernels"]["roe_march"]
            if base:
                full["static_issue_cycles"] = full["issue_cycles"]
                full["static_valu"] = full["valu"]
                full["issue_cycles"] = base["issue_cycles"] + (full["issue_cycles"] - base["issue_cycles"]) // 4
                full["valu"] = base["valu"] + (full["valu"] - base["valu"]) // 4
                full["what"] += " (average step: the fifth-j-face block counted once in four planes)"
        # registers / scratch / LDS of the Roe march as the library ships it (with the clamp block)
        o = os.path.join(td, "roe_shipped.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only",
                            os.path.join(ROOT, "adflow_amd", "csrc", "kernels_roe_march.hip"), "-o", o], capture_output=True, text=True)
        if r.returncode == 0 and "roe_march" in out["kernels"]:
            shipped = main_loop(o, KERNELS["roe_march"][1])
            if shipped:
                for key in ("NumVgprs", "ScratchSize", "LDSByteSize", "Occupancy", "NumSgprs"):
                    if key in shipped:
                        out["kernels"]["roe_march"][key] = shipped[key]
                out["kernels"]["roe_march"]["what"] += "; counted without the block a lane enters where a difference lies inside the limiter's clamp"
    p = os.path.join(ROOT, "profiles", "isa_counts.json")
    json.dump(out, open(p, "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:16s} valu {v['valu']:5d} f64 {v['f64']:5d} trans {v['trans64']:3d} loads {v['gload']:3d} lds {v['ds']:3d} scratch {v['scratch']:3d} "
              f"issue cycles {v['issue_cycles']:6d} vgpr {v.get('NumVgprs')} scratch B {v.get('ScratchSize')}")


if __name__ == "__main__":
    main()
