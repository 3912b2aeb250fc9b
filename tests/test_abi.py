"""CPU-only checks of the drop-in boundary: the HIP library loads, exports every
symbol include/adflow_gpu.h declares, its structs agree with the ctypes and the
Fortran ISO_C_BINDING mirrors, and it refuses to compute without a GPU."""
import ctypes
import os
import re

import pytest

from adflow_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "adflow_gpu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(adflow_gpu_\w+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    syms = header_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    # and the Python binding knows each of them
    assert set(syms) == set(capi.EXPORTS), set(syms) ^ set(capi.EXPORTS)


def test_struct_layouts_agree():
    lib = capi.load()
    so, sd = ctypes.c_int(), ctypes.c_int()
    assert lib.adflow_gpu_abi_sizes(ctypes.byref(so), ctypes.byref(sd)) == 0
    assert so.value == ctypes.sizeof(capi.AdflowOpts)
    assert sd.value == ctypes.sizeof(capi.AdflowBlockDesc)
    assert lib.adflow_gpu_abi_sizes2(ctypes.byref(so), ctypes.byref(sd)) == 0
    assert so.value == ctypes.sizeof(capi.AdflowBcSubface)
    assert sd.value == ctypes.sizeof(capi.AdflowCommPattern)


def test_fortran_shim_mirrors_agree():
    """adflow_amd/fortran/adflow_gpu_shim.F90 compiled against the reference's
    modules (inside oracle/_ref) reports the same struct sizes."""
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref not built")
    lib = ref.load()
    a, b = ctypes.c_int(), ctypes.c_int()
    lib.ref_shim_sizes(ctypes.byref(a), ctypes.byref(b))
    assert (a.value, b.value) == (ctypes.sizeof(capi.AdflowOpts), ctypes.sizeof(capi.AdflowBlockDesc))
    lib.ref_shim_sizes2(ctypes.byref(a), ctypes.byref(b))
    assert (a.value, b.value) == (ctypes.sizeof(capi.AdflowBcSubface), ctypes.sizeof(capi.AdflowCommPattern))


def test_no_cpu_fallback():
    """Without a visible GPU the engine must fail loudly, not compute."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from adflow_amd import capi\n"
            "lib = capi.load()\n"
            "rc = lib.adflow_gpu_init(0)\n"
            "print('rc', rc, lib.adflow_gpu_last_error().decode())\n"
            "rc2 = lib.adflow_gpu_block_res(1, 7)\n"
            "print('rc2', rc2)\n" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout
    assert "rc 1" in out and "no CPU fallback" in out, out
    assert "rc2 1" in out, out


def test_unsupported_configurations_are_refused():
    """adflow_opts::unsupported (unsteady / time spectral, cp curve fits, wall functions, overset) and sps > 1 must be
    refused instead of silently computed as steady, constant-gamma, 1-to-1 (round-1 advisor finding)."""
    import ctypes
    from adflow_amd.params import FlowParams
    lib = capi.load()
    for bit, word in ((1, "unsteady"), (2, "cpModel"), (4, "wall functions"), (8, "overset")):
        o = capi.opts_from_params(FlowParams(unsupported=bit))
        assert lib.adflow_gpu_set_options(ctypes.byref(o)) != 0
        assert word in lib.adflow_gpu_last_error().decode()
    o = capi.opts_from_params(FlowParams())
    assert lib.adflow_gpu_set_options(ctypes.byref(o)) == 0


def test_shim_hands_no_aliased_arrays_to_coarse_levels():
    """gpuRegisterBlock: gamma, rlv, dw, fw, dtl, radI/J/K alias the finest level's storage on coarse levels (setPointers)
    and must not be handed over with a coarse shape (round-1 advisor finding)."""
    src = open(os.path.join(ROOT, "adflow_amd", "fortran", "adflow_gpu_shim.F90")).read()
    blk = src[src.index("subroutine gpuRegisterBlock"):src.index("end subroutine gpuRegisterBlock")]
    for name in ("gamma", "rlv", "dw", "fw", "dtl", "radI", "radJ", "radK"):
        assert re.search(r"d%%%s = c_null_ptr" % name, blk), name
        m = re.search(r"if \(level == 1\) then(.*?)end if", blk, flags=re.S)
        assert m
    assert "sps = 1" in open(os.path.join(ROOT, "adflow_amd", "csrc", "api.hip")).read()


def test_comm_info_without_a_communicator():
    """adflow_gpu_comm_info before adflow_gpu_comm_init: rank 0 of 1, no communicator (-1, -1); needs no GPU"""
    lib = capi.load()
    v = [ctypes.c_int(7) for _ in range(4)]
    assert lib.adflow_gpu_comm_info(*[ctypes.byref(x) for x in v]) == 0
    assert [x.value for x in v] == [0, 1, -1, -1]
