import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_addoption(parser):
    parser.addoption("--hostsim", action="store_true", default=False,
                     help="developer aid: run the -m gpu tests against the host emulation of the kernels")


@pytest.fixture(scope="session", autouse=True)
def _fresh_hip_library():
    """(Re)build libadflow_gpu.so when its sources are newer (hipcc cross-compiles
    without a GPU) so the ABI tests never look at a stale library."""
    import shutil
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        from adflow_amd.build import build_lib
        build_lib(verbose=False)
    yield


@pytest.fixture(scope="session")
def engine(request):
    """The HIP engine through the C-ABI of include/adflow_gpu.h (real MI355X)."""
    from adflow_amd.engine import Engine
    if request.config.getoption("--hostsim"):
        from hostsim.build import build
        eng = Engine(0, _lib_path=build())
    else:
        eng = Engine(0)
    yield eng
    eng.close()


@pytest.fixture(scope="session")
def hostsim_engine():
    """Kernel-logic emulator (tests/hostsim): the same kernel sources compiled
    with g++; CPU-only CI coverage of the kernel arithmetic.  Never timed."""
    from adflow_amd.engine import Engine
    from hostsim.build import build
    eng = Engine(0, _lib_path=build())
    yield eng
    eng.close()


def pytest_terminal_summary(terminalreporter):
    """the largest LOCAL error |a-b| / (|b| + 1e-6 max|b|) any residual comparison of the session saw (tests/util.py)"""
    try:
        from util import worst_local_error, LOCAL_TOL
        terminalreporter.write_line(f"worst local residual error of the session: {worst_local_error():.3e} (bound {LOCAL_TOL:.0e})")
    except Exception:
        pass
