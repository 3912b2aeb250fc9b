"""Worker of tests/test_multirank_cpu.py: one rank of an N-rank halo exchange with
the inter-rank transport done by torch.distributed (gloo) between the library's
pack and unpack entry points.  The kernel side runs on the tests/hostsim
emulator (CPU-only CI); the product transport (RCCL inside
adflow_gpu_halo_exchange) replaces exactly the send/recv calls below."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from adflow_amd.engine import Engine  # noqa: E402
from adflow_amd.params import FlowParams, RANSEquations  # noqa: E402
from adflow_amd.synth import make_block  # noqa: E402
from adflow_amd.topology import BrickTopology, apply_local_copies_fast, ell_topology  # noqa: E402
from hostsim.build import build  # noqa: E402
import ctypes  # noqa: E402


def main():
    rank, world, port, nLayers = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    mode = sys.argv[5] if len(sys.argv) > 5 else "mod"
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    prm = FlowParams(equations=RANSEquations)
    dims = (6, 5, 4)
    periodic = (True, True, True)
    if mode == "strong":      # bench.py --scaling strong: ONE 2x2x2 brick, nb / N blocks per rank, contiguous in k
        shape = (2, 2, 2)
        import bench
        topo = BrickTopology(*shape, *dims, owner=bench.strong_owner(8, world))
    elif mode == "wall":      # bench.py's default workload at N = 2 (weak): 2 x 1 x 1 ranks, each a 2x2x2 brick, the ends of the whole
        shape = (4, 2, 2)     # 4x2x2 brick physical boundaries (no pattern entry beyond them), 1-to-1 interfaces inside
        periodic = (False, False, False)
        topo = BrickTopology(*shape, *dims, owner=lambda g: (g % 4) // 2, periodic=periodic)
    elif mode in ("weak", "weakwall"):
        # bench.py's weak layout at any N (Job.__init__): the ranks form an rx x ry x rz grid (rank_grid: 8 -> 2x2x2), every rank owns a
        # 2x2x2 brick of blocks; "weak": periodic in all three directions (the periodic twin), "weakwall": the ends of the whole brick are
        # physical boundaries (the default, wall-bounded workload).  At N = 8 a rank has three face peers and edge / corner peers: 7
        # (bench.py's own rank_grid / weak_owner: the layout under test IS the driver's)
        import bench
        rx, ry, rz = bench.rank_grid(world)
        e = 2
        dims = (4, 3, 3)
        shape = (e * rx, e * ry, e * rz)
        if mode == "weakwall":
            periodic = (False, False, False)
        topo = BrickTopology(*shape, *dims, owner=bench.weak_owner(e, rx, ry), periodic=periodic)
    else:
        shape = (2, 2, 1)
        topo = BrickTopology(*shape, *dims, owner=lambda g: g % world)
    if mode == "ell":         # three blocks of different sizes joined with rotated index systems (tests/test_gpu_topology.py): blocks A and
        topo = ell_topology(owner=lambda g: 0 if g != 1 else 1)       # C on rank 0, B on rank 1; B's edge halos come from C
        single = ell_topology()
        allb = {g: single.make_block(g, prm, seed=50 + g) for g in range(topo.nblocks)}
        bdims = {g: topo.dims(g) for g in range(topo.nblocks)}
    else:
        allb = {g: make_block(*dims, prm, seed=50 + g, stretch_k=2.0) for g in range(topo.nblocks)}
        single = BrickTopology(*shape, *dims, periodic=periodic)
        bdims = {g: dims for g in range(topo.nblocks)}
    lid = topo.local_ids()
    # every rank can rebuild every block (seeded): expected halos come from the
    # single-rank version of the same topology
    exp = {single.local_ids()[g]: allb[g].copy() for g in range(topo.nblocks)}
    apply_local_copies_fast(exp, single.patterns(nLayers)[0])
    mine = {lid[g]: allb[g] for g in topo.blocks_of(rank)}
    eng = Engine(0, _lib_path=build())
    eng.set_options(prm)
    for nn, b in mine.items():
        eng.register(b, nn=nn, level=1)
    cp = topo.patterns(nLayers)[rank]
    eng.comm_register(1, nLayers, cp)
    lib = eng.lib
    nvar = 6 + 1 + 2   # w(1:6), p, rlv, rev
    args = (1, nLayers)
    var = (1, 6, 1, 1)
    # same-process copies
    assert lib.adflow_gpu_halo_local_copy(1, nLayers, *var) == 0
    # inter-process: pack -> gloo isend / irecv -> unpack
    reqs, rbufs = [], []
    for s in range(len(cp.sendProc)):
        peer, cnt = ctypes.c_int(), ctypes.c_int()
        lib.adflow_gpu_halo_slot_info(1, nLayers, 1, s, ctypes.byref(peer), ctypes.byref(cnt))
        buf = torch.empty(nvar * cnt.value, dtype=torch.float64)
        assert lib.adflow_gpu_halo_pack(1, nLayers, s, *var, buf.data_ptr()) == 0
        reqs.append(dist.isend(buf, dst=peer.value))
    for r in range(len(cp.recvProc)):
        peer, cnt = ctypes.c_int(), ctypes.c_int()
        lib.adflow_gpu_halo_slot_info(1, nLayers, 0, r, ctypes.byref(peer), ctypes.byref(cnt))
        buf = torch.empty(nvar * cnt.value, dtype=torch.float64)
        rbufs.append((r, buf, dist.irecv(buf, src=peer.value)))
    for q in reqs:
        q.wait()
    for r, buf, q in rbufs:
        q.wait()
        assert lib.adflow_gpu_halo_unpack(1, nLayers, r, *var, buf.data_ptr()) == 0
    # compare with the single-rank expectation
    bad = 0
    lo = 2 - nLayers
    for g in topo.blocks_of(rank):
        nn = lid[g]
        eng.download_state(nn, 1)
        e = exp[single.local_ids()[g]]
        sl = tuple(slice(lo, n + 2 + nLayers) for n in bdims[g])
        for name in ("w", "p", "rlv", "rev"):
            a, b = mine[nn][name][sl], e[name][sl]
            if not np.array_equal(a, b):
                bad += 1
                print(f"rank {rank} block {nn} {name}: mismatch {np.abs(a - b).max()}")
    dist.barrier()
    eng.close()
    dist.destroy_process_group()
    print(f"rank {rank} peers {len(cp.sendProc)}")
    print(f"rank {rank} OK" if bad == 0 else f"rank {rank} FAIL")
    sys.exit(0 if bad == 0 else 1)


if __name__ == "__main__":
    main()
