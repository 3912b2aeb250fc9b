"""The ISO_C_BINDING host side of the boundary (adflow_amd/fortran/adflow_gpu_shim.F90) EXECUTED: Fortran code compiled
against the reference's own modules fills adflow_opts / adflow_block_desc / adflow_comm_pattern / adflow_bc_subface from
flowDoms, communication.F90 and BCData and drives the C-ABI library (tests/shim_worker.py, oracle ref_shim_roundtrip);
the residual it brings back must equal the reference's.  CPU CI: against the emulator build; -m gpu: against the HIP library."""
import os
import subprocess
import sys

import pytest

from oracle import ref

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built")

CASES = [("brick", 3, 9), ("brick", 1, 1), ("bocos", 3, 9), ("bocos", 2, 2)]


def run(kind, case, eq, sd):
    r = subprocess.run([sys.executable, os.path.join(HERE, "shim_worker.py"), kind, case, str(eq), str(sd)], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0 and "SHIM OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("case,eq,sd", CASES)
def test_fortran_shim_executes_on_emulator(case, eq, sd):
    run("hostsim", case, eq, sd)


@pytest.mark.gpu
@pytest.mark.parametrize("case,eq,sd", CASES)
def test_fortran_shim_executes_on_gpu(case, eq, sd):
    run("hip", case, eq, sd)
