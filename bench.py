#!/usr/bin/env python
"""Benchmark of the residual-evaluation hot path on MI355X.

`python bench.py --gpus N --steps K --warmup W` — one "step" is ONE residual
evaluation (blockette::blocketteRes core: time step + inviscid [+ viscous + SA]
+ final sum) over every block of the workload, state resident in HBM.

metric  : Mcells*residual-evals/s  (BASELINE.json)
roofline: HBM-bound; algorithmic bytes/cell/eval from SURVEY.md §8(d)
          (Euler 175 B, RANS-SA 255 B) over the live HIP-event duration of the
          dominant kernel on the library's own stream.
cpu_baseline: the reference's own Fortran (oracle/_ref, "reference") timed on
          the host cores of this box on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec

WORKLOADS = {
    # BASELINE.json configs[1]: tutorial-wing multiblock Euler, JST scalar, roofline size
    "euler_jst_8x128": dict(equations=1, spaceDiscr=1, nblocks=8, dims=(128, 128, 128), bytes_per_cell=175.0),
    # BASELINE.json configs[2]/[3] at roofline size (not the headline line; `--workload ...`):
    # RANS + SA, viscous flux + SA residual, scalar JST / Roe upwind / matrix dissipation
    "rans_sa_jst_8x128x128x96": dict(equations=3, spaceDiscr=1, nblocks=8, dims=(128, 128, 96), bytes_per_cell=255.0),
    "rans_sa_upwind_8x128x128x96": dict(equations=3, spaceDiscr=9, nblocks=8, dims=(128, 128, 96), bytes_per_cell=255.0),
    "rans_sa_matrix_8x128x128x96": dict(equations=3, spaceDiscr=2, nblocks=8, dims=(128, 128, 96), bytes_per_cell=255.0),
    # same cell count as the headline, cut into 512 blocks of 32^3 (what a production multiblock mesh looks like per GPU):
    # measures how much of the rate survives small blocks (level-batched launches)
    "euler_jst_512x32": dict(equations=1, spaceDiscr=1, nblocks=512, dims=(32, 32, 32), bytes_per_cell=175.0),
}


CPU_WORKER = r"""
import sys, json
sys.path.insert(0, sys.argv[1])
from adflow_amd.params import FlowParams
from adflow_amd.synth import make_block
from oracle import ref
n1, n2, n3, equations, seconds, seed = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]), int(sys.argv[7])
prm = FlowParams(equations=equations)
blk = make_block(n1, n2, n3, prm, seed=seed)
ref.bind_block(blk, prm)
n, dt = ref.time_block_res_core(seconds, True, True, equations == 3)
print(json.dumps({"rate": blk.ncells * n / dt}))
"""


def cpu_baseline(equations, seconds=12.0, max_cores=32, dims=(64, 64, 64)):
    """The reference's own Fortran (oracle/_ref) on this box's host cores: one
    pinned process per core, each repeating blockResCore on its own block."""
    import subprocess
    from oracle import ref
    if not ref.available():
        return None
    avail = sorted(os.sched_getaffinity(0))
    cores = max(1, min(len(avail), max_cores))
    procs = []
    for i in range(cores):
        cmd = ["taskset", "-c", str(avail[i]), sys.executable, "-c", CPU_WORKER, ROOT,
               str(dims[0]), str(dims[1]), str(dims[2]), str(equations), str(seconds), str(100 + i)]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
    rates = []
    deadline = time.time() + seconds + 120.0
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=max(1.0, deadline - time.time()))
            rates.append(json.loads(out.strip().splitlines()[-1])["rate"])
        except Exception:
            pr.kill()
    if not rates:
        return None
    return {"value": sum(rates) / 1e6, "unit": "Mcells*residual-evals/s", "cores": len(rates), "kind": "reference",
            "sample": f"{len(rates)} pinned processes x one {dims[0]}x{dims[1]}x{dims[2]} block each, ~{seconds:.0f} s of "
                      "blockResCore evaluations of the reference Fortran (amdflang -O3 -fdefault-real-8); no MPI halo exchange",
            "per_core": sum(rates) / len(rates) / 1e6}


T_START = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="euler_jst_8x128")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mg", action="store_true", help="skip the MG-cycles/s measurement")
    ap.add_argument("--tuning", action="append", default=[], help="key=value knobs of adflow_gpu_set_tuning")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    log("torch imported")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)

    from adflow_amd.engine import Engine
    from adflow_amd.params import FlowParams
    from adflow_amd.synth import make_block

    wl = WORKLOADS[a.workload]
    prm = FlowParams(equations=wl["equations"], spaceDiscr=wl["spaceDiscr"],
                     vis4=0.1 if wl["spaceDiscr"] == 2 else 0.0156)
    eng = Engine(local_rank)
    eng.set_options(prm)
    tuning = dict(kv.split("=") for kv in a.tuning)
    for k_, v_ in tuning.items():
        eng.set_tuning(k_, int(v_))
    march = int(tuning.get("euler_march", 1)) and wl["equations"] == 1 and wl["spaceDiscr"] == 1
    # weak scaling: every GPU owns `nblocks` blocks (a 2x2x2 brick) of the workload; the
    # ranks' bricks are chained along i into one periodic brick of (2N)x2x2 blocks, so
    # every evaluation is preceded by the 2-layer halo exchange the reference's
    # blocketteRes performs (whalo2, blockette.F90:246): same-GPU copies + RCCL p2p
    from adflow_amd.topology import BrickTopology
    nb = wl["nblocks"]
    dims = wl["dims"]
    e = round(nb ** (1.0 / 3.0))                 # per-GPU brick of e x e x e blocks
    assert e ** 3 == nb
    topo = BrickTopology(e * world, e, e, *dims, owner=lambda g: (g % (e * world)) // e)
    lid = topo.local_ids()
    cells_local = 0
    from adflow_amd.synth import make_coarse_block
    for g in topo.blocks_of(rank):
        blk = make_block(*dims, prm, seed=20260925 + g, stretch_k=3.0 if wl["equations"] == 3 else 1.0)
        both = [blk]
        if not a.no_mg and wl["equations"] != 3:     # RANS workloads time a single-grid iteration (config 3)
            cblk = make_coarse_block(blk, prm, seed=777 + g)   # also attaches mgI/J/KCoarse to blk
            both.append(cblk)
        eng.register(blk, nn=lid[g], level=1)
        if len(both) > 1:
            eng.register(cblk, nn=lid[g], level=2)
        cells_local += blk.ncells
        log(f"block {lid[g]}/{nb} generated and uploaded")
        # host copies are no longer needed by the timed loop
        for b_ in both:
            for k in list(b_.a.keys()):
                if k not in ("dw",):
                    del b_.a[k]
    halo = "off"
    try:
        cp = topo.patterns(2, only_rank=rank)[rank]
        eng.comm_register(1, 2, cp)
        log(f"comm pattern: {cp.ncopy} local copies, {int(cp.nsendCum[-1])} cells sent to {cp.sendProc.size} ranks")
        # RCCL communicator of the library (also at N=1: exercises the bootstrap)
        idbuf = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            import ctypes
            raw = (ctypes.c_char * 128)()
            from adflow_amd import capi
            capi.check(eng.lib.adflow_gpu_comm_unique_id(raw), eng.lib)
            idbuf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
        if world > 1:
            idg = idbuf.cuda()
            dist.broadcast(idg, 0)
            idbuf = idg.cpu()
        from adflow_amd import capi
        capi.check(eng.lib.adflow_gpu_comm_init(rank, world, idbuf.numpy().tobytes()), eng.lib)
        eng.whalo2(1, 1, prm.nw)
        halo = "whalo2 every step: same-GPU copies" + (" + RCCL send/recv over xGMI" if world > 1 else "")
    except Exception as e:  # the evaluation of independent shards is still a valid measurement
        halo = f"FAILED ({e}); shards evaluated without exchange"
        log("halo exchange unavailable: " + str(e))
    do_halo = not halo.startswith("FAILED")

    def step():
        if do_halo:
            eng.whalo2(1, 1, prm.nw)
        eng.blocketteRes(1, True, True, wl["equations"] == 3)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    eng.set_async(True)
    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    eng.event_record(0)
    for _ in range(a.steps):
        step()
    eng.event_record(1)
    barrier()
    dt = time.perf_counter() - t0
    ev_ms = eng.event_elapsed_ms(0, 1)
    log(f"timed loop done: {dt / a.steps * 1e3:.3f} ms/step")

    # dominant kernel alone (flux kernel: `residual` entry = initres+residual_block), live HIP events
    from adflow_amd.params import DADI
    eng.set_options(prm.replace(smoother=DADI))   # rFil = 1, non-persistent fw: the variant blocketteRes launches
    for _ in range(2):
        eng.residual(1, 0)
    eng.event_record(2)
    for _ in range(a.steps):
        eng.residual(1, 0)
    eng.event_record(3)
    eng.sync()
    # the k-marching Euler kernel covers all blocks of the level in ONE launch
    launches_per_step = 1 if march else nb
    k_ms = eng.event_elapsed_ms(2, 3) / (a.steps * launches_per_step)
    eng.set_async(False)

    # ---- second headline metric: multigrid cycles / s (2-level V cycle, RK smoother) ----
    mg = None
    if not a.no_mg:
        try:
            from adflow_amd.params import RungeKutta, DADI, alternateResAveraging, noResAveraging
            rans = wl["equations"] == 3
            if rans:
                # BASELINE config 3 (test_functionals.py:136-160): single grid, D-ADI with 3 sub-iterations,
                # 3 SA DDADI sub-iterations, cfl 1.5, no residual averaging: one "cycle" = one solver iteration
                eng.set_options(prm.replace(smoother=DADI, nSubiterations=3, nSubIterTurb=3, cfl=1.5, resAveraging=noResAveraging))
                cyc_desc = "single grid: D-ADI x3 sub-iterations + SA DDADI x3 (BASELINE config 3)"
            else:
                # pyADflow defaults (pyADflow.py:5697-5731): RK smoother, "alternate" residual averaging
                eng.set_options(prm.replace(smoother=RungeKutta, resAveraging=alternateResAveraging))
                cyc_desc = "2-level V: smooth(RK5, alternate residual averaging) / restrict / smooth / prolong + closing residual"
            ctopo = BrickTopology(e * world, e, e, dims[0] // 2, dims[1] // 2, dims[2] // 2, owner=topo.owner)
            if do_halo:
                eng.comm_register(1, 2, cp)
                if not rans:
                    eng.comm_register(2, 1, ctopo.patterns(1, only_rank=rank)[rank])
            log("multigrid levels registered")
            cyc = [0] if rans else [0, 1, 0, -1]
            eng.set_async(False)
            eng.timeStep(1, False)
            eng.residual(1, 0)
            for _ in range(2):
                eng.executeMGCycle(cyc)
            barrier()
            ncyc = max(5, a.steps // 5)
            t1 = time.perf_counter()
            for _ in range(ncyc):
                eng.executeMGCycle(cyc)
            barrier()
            dt_mg = time.perf_counter() - t1
            if world > 1:
                tt = torch.tensor([dt_mg], dtype=torch.float64, device="cuda")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt_mg = float(tt.item())
            # algorithmic bytes of one cycle: sum over levels of n_smooth x nStages x (B_res + B_upd)
            # + the residuals of the transfers + restriction / prolongation, level l holding N / 8^l cells
            if rans:
                b_res, b_dadi, b_upd = 255.0 + 32.0, 244.0, 224.0
                per_cell = 3 * (b_res + b_dadi + b_upd)            # SA DDADI sweeps not in the §8(d) table: left out
                formula = "3 x (B_res 287 + B_dadi 244 + B_upd 224) B per cell, SA solve not counted"
            else:
                b_res, b_upd, b_ts, b_tr = 175.0, 200.0, 32.0, 8.0 * (2 * 5 + 2)
                smooth = 5 * (b_res + b_upd)
                per_cell = smooth + (b_res + b_ts) + (b_res + smooth) / 8.0 + b_tr + (b_res + b_ts)
                formula = ("fine RK5 5 x (175 + 200) + restriction residual (175 + 32) + coarse [forcing residual 175 + RK5] / 8 "
                           "+ transfers 96 + closing residual (175 + 32) B per fine cell")
            alg_cycle = per_cell * cells_local * world
            mg = {"cycles_per_s": ncyc / dt_mg, "ms_per_cycle": dt_mg / ncyc * 1e3, "cycles_timed": ncyc,
                  "cycle": cyc_desc,
                  "fine_cells_per_gpu": cells_local,
                  "algorithmic_bytes_per_cycle": alg_cycle, "algorithmic_bytes_formula": formula,
                  "hbm_frac": alg_cycle / (dt_mg / ncyc) / (8.0e12 * world)}
            log(f"MG: {mg['ms_per_cycle']:.3f} ms/cycle")
        except Exception as e:
            mg = {"error": str(e)}
            log("MG benchmark failed: " + str(e))

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    cells_total = cells_local * world
    value = cells_total * a.steps / dt / 1e6

    out = None
    if rank == 0:
        # HBM-side bytes per launch of the dominant kernel from the PMC passes over this very command
        # (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc runs, gfx950 correction of
        # MI355X_MICROARCH.md applied; tools/pmc_traffic.py writes the file, the raw summary sits beside it)
        traffic, traffic_src = None, None
        try:
            tf = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            ent = tf.get(a.workload)
            if ent and not a.tuning:
                traffic, traffic_src = ent["traffic_bytes_per_launch"], ent["source"]
        except (OSError, ValueError, KeyError):
            pass
        cells_per_launch = cells_local / launches_per_step
        alg_bytes = wl["bytes_per_cell"] * cells_per_launch
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "Mcells*residual-evals/s", "value": value, "unit": "Mcells*residual-evals/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{a.workload}: {nb} blocks x {wl['dims'][0]}x{wl['dims'][1]}x{wl['dims'][2]} cells per GPU, "
                                   + ("Euler, central + scalar JST" if wl["equations"] == 1 else "RANS-SA, spaceDiscr=%d" % wl["spaceDiscr"])
                                   + ", one residual evaluation per step (whalo2 + blocketteRes core)",
                       "halo_exchange": halo,
                       "cells_per_gpu": cells_local, "device": eng.device_name()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_euler_march_p" if march else ("k_inviscid" if wl["equations"] == 1 else
                                                                   "k_inviscid + k_nodal_gradients + k_viscous (one block)"),
                         "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": alg_bytes},
            "mg": mg,
            "whole_eval": {"event_ms_per_step": ev_ms / a.steps,
                           "hbm_frac": wl["bytes_per_cell"] * cells_local / (ev_ms / a.steps * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
    eng.close()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            log("cpu baseline (reference Fortran on the host cores) ...")
            out["cpu_baseline"] = cpu_baseline(wl["equations"])
            log("cpu baseline done")
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    sys.stdout.flush()
    sys.stderr.flush()
    # RCCL prints a banner through buffered C stdio that is flushed at process exit:
    # point fd 1 at /dev/null from here on so the JSON line stays the last line of
    # stdout (normal exit still runs atexit handlers, e.g. rocprofv3's writer)
    dn = os.open(os.devnull, os.O_WRONLY)
    os.dup2(dn, 1)


if __name__ == "__main__":
    main()
