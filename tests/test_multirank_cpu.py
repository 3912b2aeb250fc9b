"""CPU-only, world_size 2 (gloo): the N>1 halo-exchange path — pattern split
over ranks, pack, transport, unpack, same-process copies — against the
single-rank result.  Kernels run on the tests/hostsim emulator."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("nLayers,mode", [(1, "mod"), (2, "mod"), (2, "strong"), (2, "wall"), (1, "ell"), (2, "ell")])
def test_two_rank_halo_exchange_gloo(nLayers, mode):
    """mode "strong": the partition of bench.py --scaling strong (one 2x2x2 brick, 8 / N blocks per rank); "wall": the weak-scaling
    layout of the default (wall-bounded) workload at two ranks: a non-periodic 4x2x2 brick, one 2x2x2 half per rank; "ell": three
    blocks of different sizes whose interfaces carry a transformation (one of them, with an index running against its neighbour's,
    on the other rank)"""
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mp_halo_worker.py"), str(r), "2", str(port), str(nLayers), mode],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            o = "TIMEOUT"
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {r} OK" in o, o[-2000:]
