#!/usr/bin/env python
"""Benchmark of the residual-evaluation hot path on MI355X.

`python bench.py --gpus N --steps K --warmup W` -- one "step" is ONE residual
evaluation over every block of the workload, state resident in HBM:
whalo2 (2-layer ghost-cell exchange) + the blocketteRes core with its default
flags (blockette.F90:118-140: exact residual, flow + turbulence, no
intermediate update) -- what the NK matrix-free matvec, getResidual and the
adjoint call.

Default workload (BASELINE.json `metric` / configs[3]):
CRM wing-body RANS-SA at roofline size, 8 blocks x 160x128x64 per GPU, Roe
upwind with the van Albada limiter (kappa = 1/3), as a WALL-BOUNDED 2x2x2 brick
(round 4: viscous wall below the four lower blocks, one symmetry plane,
farfield elsewhere, 1-to-1 interfaces inside), stepped as the reference's WHOLE
blocketteRes (blockette.F90:199-283): derived values, turbulence + mean-flow
boundary conditions, whalo2, core with storeWall.  Other configurations are
reported under "extra" at N=1 (the fully periodic brick of rounds 1-3, 4b matrix
dissipation, config 5 GMRES proxy, config 3 D-ADI iteration, config 2 Euler JST
+ 3-level W multigrid cycle, small blocks).

metric  : Mcells*residual-evals/s  (BASELINE.json)
roofline: HBM-bound; algorithmic bytes/cell/eval from SURVEY.md §8(d)
          (Euler 175 B, RANS-SA 255 B).  Per-kernel durations are measured live
          with HIP events on the library's own stream (tuning "phase_events").
cpu_baseline: the reference's own Fortran (oracle/_ref, "reference") timed on
          the host cores of this box on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
# context figures (never the priced peak): the streaming ceiling and the sustained FP64 issue clock, measured on a box of the pool by
# `tools/pmc_calib.bin bw2` (16 B per lane, 8 loads in flight per thread, exact grid; v_fma_f64 chains) -> profiles/calibration.json;
# without that file the guide's figures: 6.29 TB/s (float4 copy), 2.4 GHz
HBM_STREAM_GBS = 6290.0
ISSUE_CLOCK_GHZ = 2.4
N_SIMD_PER_CU = 4


def load_calibration():
    try:
        c = json.load(open(os.path.join(ROOT, "profiles", "calibration.json")))
        return c
    except (OSError, ValueError):
        return {}


def fp64_issue_roofline(eng, eval_ms, kernels_ms, qcr=False, gf=True, spaceDiscr=9):
    """FP64 VALU issue bound of the evaluation: wavefront-steps of every marching kernel (adflow_gpu_march_stats) x the issue cycles of
    its main loop (profiles/isa_counts.json: 4 cycles per VALU instruction of a 64-wide wavefront, 16 per f64 transcendental seed) over
    all SIMDs at the sustained issue clock.  The kernels share the SIMDs, so the floor of the evaluation is the SUM."""
    try:
        isa = json.load(open(os.path.join(ROOT, "profiles", "isa_counts.json")))
        steps = eng.march_stats(1)
        name = eng.device_name()
        cus = int(name.split(",")[-1].split("CUs")[0]) if "CUs" in name else 256
    except Exception as ex:   # no counts committed: no issue roofline
        return {"error": str(ex)}
    cal = load_calibration()
    clock = float(cal.get("issue_clock_ghz_at_4_cycles", ISSUE_CLOCK_GHZ))
    simds = cus * N_SIMD_PER_CU
    inv = {9: "roe_march", 2: "matrix_march", 1: "euler_march"}[spaceDiscr]
    plan = [("SA residual", "sa_march", "sa_march"), ("inviscid", inv, "tile_march")]
    plan.append(("nodal gradients + viscous (fused)", "visc_gf_qcr" if qcr else "visc_gf", "visc_gf"))
    # dynamic instruction counts (SQ_INSTS_VALU per launch, tools/_gpu_job_sq.sh -> profiles/pmc_sq.json) where they were collected:
    # the assembly holds both sides of every uniform branch (normals from the nodes or from the arrays, first / second order, the face
    # part wave 0 of k_visc_gf skips), the counter only what was issued
    try:
        sq = json.load(open(os.path.join(ROOT, "profiles", "pmc_sq.json"))).get(DEFAULT_WORKLOAD, {})
    except (OSError, ValueError):
        sq = {}
    kern, total, total_dyn, have_dyn = {}, 0.0, 0.0, bool(sq.get("kernels")) and spaceDiscr == 9 and not qcr
    for label, ik, sk in plan:
        k = isa["kernels"].get(ik)
        if not k:
            continue
        ms = steps[sk] * k["issue_cycles"] / (simds * clock * 1e9) * 1e3
        kern[label] = {"kernel": k["what"], "wave_steps": steps[sk], "valu_per_step": k["valu"], "f64_per_step": k["f64"],
                       "issue_cycles_per_step": k["issue_cycles"], "issue_floor_ms": ms,
                       "measured_ms": kernels_ms.get(label), "frac_of_issue": (ms / kernels_ms[label]) if kernels_ms.get(label) else None}
        total += ms
        dyn = (sq.get("kernels", {}).get(label) or {}).get("SQ_INSTS_VALU") if have_dyn else None
        if dyn:
            msd = dyn * (k["issue_cycles"] / k["valu"]) / (simds * clock * 1e9) * 1e3
            kern[label].update({"insts_valu_counted": dyn, "issue_floor_ms_counted": msd,
                                "frac_of_issue_counted": (msd / kernels_ms[label]) if kernels_ms.get(label) else None,
                                "valu_active_per_wave_cycle": sq["kernels"][label].get("valu_active_per_wave_cycle"),
                                "wait_any_per_wave_cycle": sq["kernels"][label].get("wait_any_per_wave_cycle")})
            total_dyn += msd
        else:
            have_dyn = False
    return {"bound": "fp64 valu issue", "floor_ms": total, "frac": total / eval_ms if eval_ms else None,
            "floor_ms_counted": total_dyn if have_dyn else None, "frac_counted": (total_dyn / eval_ms) if have_dyn and eval_ms else None,
            "counted_source": sq.get("source") if have_dyn else None, "issue_clock_ghz": clock,
            "issue_clock_source": "profiles/calibration.json (tools/pmc_calib.bin bw2)" if "issue_clock_ghz_at_4_cycles" in cal
            else "MI355X_MICROARCH.md peak engine clock", "simds": simds, "isa_counts_git": isa.get("git"), "kernels": kern,
            "note": "floor = sum over the kernels (they share the SIMDs); frac = floor / measured evaluation; *_counted: the same from "
                    "SQ_INSTS_VALU per launch instead of the static count of the march loop (uniform branches not taken are not issued)"}

# ends of the wall-bounded brick (faceID of the brick: BCType): viscous adiabatic wall at kMin, symmetry plane at jMin, farfield elsewhere
WALL_BRICK = {1: -6, 2: -6, 3: -1, 4: -6, 5: -3, 6: -6}
WORKLOADS = {
    # BASELINE.json configs[3] (the configuration the metric is quoted on), SURVEY §8(d) rows 4a / 4b -- with physical boundaries
    # (round-3 verdict, next 1): what a wall-bounded CRM mesh executes, boundary conditions and derived values inside the step
    # algorithmic bytes: the core's 255 B per cell + the derived-values pass the whole blocketteRes starts with (reads rho,
    # u, v, w, rhoE, nuTilde = 48 B, writes p, rlv, rev = 24 B per cell: it cannot be folded into the core, the boundary conditions and
    # the exchange between them need its results); boundary conditions and halos are surface terms and are not counted
    "crm_rans_sa_upwind_8x160x128x64_bc": dict(equations=3, spaceDiscr=9, nblocks=8, dims=(160, 128, 64), bytes_per_cell=255.0, bc=WALL_BRICK,
                                               bytes_front=72.0,
                                               desc="RANS-SA, Roe upwind (van Albada, kappa=1/3); non-periodic 2x2x2 brick: viscous wall "
                                                    "(kMin of the 4 lower blocks), symmetry plane (jMin), farfield"),
    # the fully periodic brick of rounds 1-3 (no boundary subfaces; the step is whalo2 + the blocketteRes core only)
    "crm_rans_sa_upwind_8x160x128x64": dict(equations=3, spaceDiscr=9, nblocks=8, dims=(160, 128, 64), bytes_per_cell=255.0,
                                            desc="RANS-SA, Roe upwind (van Albada, kappa=1/3)"),
    "crm_rans_sa_matrix_8x160x128x64": dict(equations=3, spaceDiscr=2, nblocks=8, dims=(160, 128, 64), bytes_per_cell=255.0,
                                            desc="RANS-SA, matrix dissipation (vis4=0.1)"),
    # BASELINE.json configs[1]: tutorial-wing multiblock Euler, JST scalar, roofline size
    "euler_jst_8x128": dict(equations=1, spaceDiscr=1, nblocks=8, dims=(128, 128, 128), bytes_per_cell=175.0,
                            desc="Euler, central + scalar JST"),
    # BASELINE.md section 2, config 1 (the reference's own CPU-runnable single-block Euler case at roofline size): a thin 2-plane
    # block -- the worst case of a k-march, whose pipeline fills every two planes -- and a 192^3 block
    "euler_jst_1x512x256x2": dict(equations=1, spaceDiscr=1, nblocks=1, dims=(512, 256, 2), bytes_per_cell=175.0,
                                  desc="Euler, central + scalar JST, one block of two cell planes"),
    "euler_jst_1x192": dict(equations=1, spaceDiscr=1, nblocks=1, dims=(192, 192, 192), bytes_per_cell=175.0,
                            desc="Euler, central + scalar JST, one 192^3 block"),
    # BASELINE.json configs[2] at roofline size
    "rans_sa_jst_8x128x128x96": dict(equations=3, spaceDiscr=1, nblocks=8, dims=(128, 128, 96), bytes_per_cell=255.0,
                                     desc="RANS-SA, scalar JST"),
    "rans_sa_upwind_8x128x128x96": dict(equations=3, spaceDiscr=9, nblocks=8, dims=(128, 128, 96), bytes_per_cell=255.0,
                                        desc="RANS-SA, Roe upwind (van Albada)"),
    "rans_sa_matrix_8x128x128x96": dict(equations=3, spaceDiscr=2, nblocks=8, dims=(128, 128, 96), bytes_per_cell=255.0,
                                        desc="RANS-SA, matrix dissipation (vis4=0.1)"),
    # same cell count as the Euler workload, cut into 512 blocks of 32^3 (a production multiblock mesh per GPU)
    "euler_jst_512x32": dict(equations=1, spaceDiscr=1, nblocks=512, dims=(32, 32, 32), bytes_per_cell=175.0,
                             desc="Euler, central + scalar JST, 32^3 blocks"),
    # the shard ONE GPU holds when the 8-block mesh is split over 8 GPUs (BASELINE north_star, --scaling strong at N = 8): one periodic
    # block whose six interfaces all lead to other ranks.  At N = 1 its neighbours are itself; with tuning comm_self every interface
    # takes the inter-GPU path (pack, ncclSend / ncclRecv to the own rank, unpack): extra "strong_shard_one_block"
    "crm_strong_shard_1x160x128x64": dict(equations=3, spaceDiscr=9, nblocks=1, dims=(160, 128, 64), bytes_per_cell=255.0,
                                          desc="RANS-SA, Roe upwind: one block = the N = 8 shard of the north-star mesh"),
    # about the cell count of the north-star workload (11.2 M) in 343 blocks of 32^3
    "rans_sa_upwind_343x32": dict(equations=3, spaceDiscr=9, nblocks=343, dims=(32, 32, 32), bytes_per_cell=255.0,
                                  desc="RANS-SA, Roe upwind (van Albada), 32^3 blocks"),
}
DEFAULT_WORKLOAD = "crm_rans_sa_upwind_8x160x128x64_bc"
PERIODIC_TWIN = "crm_rans_sa_upwind_8x160x128x64"
PHASES = ["closures+bc", "time step", "SA residual", "inviscid", "nodal gradients", "viscous"]   # between marks 0..6 of api.hip
# NS / RANS over the tile table: k_visc_gf (gradients + viscous fluxes, one kernel between marks 4 and 5) runs in front of the Roe /
# matrix march, which adds its sums and completes dw
PHASES_GF_FIRST = ["closures+bc", "time step", "SA residual", "(mark)", "nodal gradients + viscous (fused)", "inviscid"]
PHASES_GF = ["closures+bc", "time step", "SA residual", "inviscid", "(mark)", "nodal gradients + viscous (fused)"]


CPU_WORKER = r"""
import sys, json
sys.path.insert(0, sys.argv[1])
from adflow_amd.params import FlowParams
from adflow_amd.synth import make_block
from oracle import ref
n1, n2, n3, equations, spaceDiscr, seconds, seed = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]),
                                                    int(sys.argv[6]), float(sys.argv[7]), int(sys.argv[8]))
path, fast = sys.argv[9], int(sys.argv[10])
ref.load(fast=bool(fast))
prm = FlowParams(equations=equations, spaceDiscr=spaceDiscr, vis4=0.1 if spaceDiscr == 2 else 0.0156)
blk = make_block(n1, n2, n3, prm, seed=seed, stretch_k=3.0 if equations == 3 else 1.0)
ref.bind_block(blk, prm)
n, dt = ref.time_block_res_core(seconds, False, True, equations == 3, blockette=(path == "blockette"))
print(json.dumps({"rate": blk.ncells * n / dt}))
"""


def host_topology():
    """What the CPU baseline runs on (round-4 verdict, weak 11 / next 9): logical CPUs in the affinity mask, their physical cores,
    sockets and NUMA nodes from sysfs, and the CPU quota of the cgroup -- 256 pinned processes that deliver the rate of 32 are either
    bound by the host's memory system or throttled by a quota; the JSON says which can apply."""
    avail = sorted(os.sched_getaffinity(0))

    def rd(path):
        try:
            return open(path).read().strip()
        except OSError:
            return None
    cores, sockets = {}, set()
    for c in avail:
        base = f"/sys/devices/system/cpu/cpu{c}/topology/"
        pk, cid = rd(base + "physical_package_id"), rd(base + "core_id")
        key = (pk, cid) if cid is not None else (None, c)
        cores.setdefault(key, []).append(c)
        sockets.add(pk)
    node_of = {}
    try:
        for d in sorted(os.listdir("/sys/devices/system/node")):
            if d.startswith("node") and d[4:].isdigit():
                for part in (rd(f"/sys/devices/system/node/{d}/cpulist") or "").split(","):
                    if not part:
                        continue
                    lo, _, hi = part.partition("-")
                    for c in range(int(lo), int(hi or lo) + 1):
                        node_of[c] = int(d[4:])
    except OSError:
        pass
    quota = rd("/sys/fs/cgroup/cpu.max")
    if quota is None:
        q, per = rd("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), rd("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
        quota = f"{q} {per}" if q is not None else None
    quota_cpus = None
    if quota and quota.split()[0] not in ("max", "-1"):
        try:
            quota_cpus = float(quota.split()[0]) / float(quota.split()[1])
        except (ValueError, IndexError, ZeroDivisionError):
            pass
    model = None
    for line in (rd("/proc/cpuinfo") or "").splitlines():
        if line.startswith("model name"):
            model = line.split(":", 1)[1].strip()
            break
    # placement order: one logical CPU per physical core first, the cores taken round-robin over the NUMA nodes; the SMT siblings after
    by_node = {}
    for key, cpus in sorted(cores.items(), key=lambda kv: kv[1][0]):
        by_node.setdefault(node_of.get(cpus[0], 0), []).append(cpus)
    first, second = [], []
    lists = [v for _, v in sorted(by_node.items())]
    for r in range(max((len(v) for v in lists), default=0)):
        for v in lists:
            if r < len(v):
                first.append(v[r][0])
                second.extend(v[r][1:])
    return {"cpu_model": model, "logical_cpus_in_mask": len(avail), "physical_cores_in_mask": len(cores), "sockets": len(sockets),
            "numa_nodes": len(set(node_of.get(c, 0) for c in avail)), "smt_threads_per_core": max((len(v) for v in cores.values()), default=1),
            "cgroup_cpu_max": quota, "cgroup_quota_cpus": quota_cpus, "placement": first + second}


def _cpu_run(equations, spaceDiscr, seconds, cores, dims, path, fast):
    """`cores` pinned processes, each repeating the reference's residual core on its own block; returns the list of rates.
    Placement: one process per PHYSICAL core, spread round-robin over the NUMA nodes; SMT siblings only beyond that."""
    avail = host_topology()["placement"]
    procs = []
    for i in range(cores):
        cmd = ["taskset", "-c", str(avail[i]), sys.executable, "-c", CPU_WORKER, ROOT,
               str(dims[0]), str(dims[1]), str(dims[2]), str(equations), str(spaceDiscr), str(seconds), str(100 + i), path, str(int(fast))]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
    rates = []
    deadline = time.time() + seconds + 120.0
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=max(1.0, deadline - time.time()))
            rates.append(json.loads(out.strip().splitlines()[-1])["rate"])
        except Exception:
            pr.kill()
    return rates


def cpu_baseline(equations, spaceDiscr, seconds=8.0, max_cores=None, dims=(64, 64, 64)):
    """The reference's own Fortran (oracle/_ref) on this box's host cores, as BASELINE.md section 2 asks: the DEFAULT residual path
    of pyADflow, blocketteResCore (blockette.F90:299-753), one pinned process per core; beside it the same with the reference's
    production flags (-ffast-math, config.LINUX_GFORTRAN.mk:34-36), a solo run on ONE core (no memory-bandwidth contention), and the
    unblocked twin blockResCore that round 1 / 2 timed."""
    from oracle import ref
    if not ref.available():
        return None
    # every core this process may run on (round-3 verdict, weak 13: no cap; the box's counts are reported beside it)
    avail = len(os.sched_getaffinity(0))
    cores = max(1, min(avail, max_cores) if max_cores else avail)
    what = {1: "Euler", 2: "laminar NS", 3: "RANS-SA"}[equations] + {1: " scalar JST", 2: " matrix dissipation", 9: " Roe upwind"}[spaceDiscr]
    main = _cpu_run(equations, spaceDiscr, seconds, cores, dims, "blockette", False)
    if not main:
        return None
    topo = host_topology()
    topo.pop("placement", None)
    out = {"value": sum(main) / 1e6, "unit": "Mcells*residual-evals/s", "cores": len(main), "kind": "reference",
           "host_logical_cpus": os.cpu_count(), "host_cpus_in_affinity_mask": avail, "host": topo,
           "placement": "one pinned process per physical core, cores taken round-robin over the NUMA nodes, SMT siblings last",
           # `cores` = pinned processes; under a cgroup quota they share that many CPUs' worth of time
           "effective_cpus": min(len(main), topo["cgroup_quota_cpus"]) if topo.get("cgroup_quota_cpus") else len(main),
           "sample": f"{len(main)} pinned processes x one {dims[0]}x{dims[1]}x{dims[2]} block each ({what}), ~{seconds:.0f} s of "
                     "blocketteResCore evaluations (blockette.F90:299-753, the default residual path; updateIntermed=F) of the reference "
                     "Fortran (amdflang -O3 -fdefault-real-8); no MPI halo exchange",
           "per_core": sum(main) / len(main) / 1e6}
    solo = _cpu_run(equations, spaceDiscr, min(seconds, 5.0), 1, dims, "blockette", False)
    if solo:
        out["solo_one_core"] = solo[0] / 1e6
    if ref.available(fast=True):
        fast = _cpu_run(equations, spaceDiscr, min(seconds, 5.0), cores, dims, "blockette", True)
        if fast:
            out["fast_math"] = {"value": sum(fast) / 1e6, "cores": len(fast), "flags": "-O3 -ffast-math (oracle/refbuild/Makefile FAST=1)"}
    if cores > 32:
        # fewer processes can deliver more: the sweep says where the rate saturates (and, with `host`, whether that is the physical
        # core count, a cgroup quota or the memory system).  `value` stays the all-core run the task asks for, `best` the largest
        sweep = {}
        for n in (8, 32, 64, 128):
            if n >= cores:
                break
            sub = _cpu_run(equations, spaceDiscr, min(seconds, 4.0), n, dims, "blockette", False)
            if sub:
                sweep[str(len(sub))] = {"value": sum(sub) / 1e6, "per_process": sum(sub) / len(sub) / 1e6}
        if solo:
            sweep["1"] = {"value": solo[0] / 1e6, "per_process": solo[0] / 1e6}
        sweep[str(len(main))] = {"value": out["value"], "per_process": out["per_core"]}
        out["sweep"] = sweep
        if "32" in sweep:
            out["cores_32"] = {"value": sweep["32"]["value"], "cores": 32}
        out["best"] = max(v["value"] for v in sweep.values())
        # what limits it: a quota below the process count throttles; else per-process rates that fall with the count while the total
        # stays flat are the memory system (the residual streams ~25 arrays of its 64^3 block per evaluation)
        q = (out["host"] or {}).get("cgroup_quota_cpus")
        out["saturation"] = ("cgroup CPU quota of %.1f CPUs" % q) if q and q < len(main) else \
            "no CPU quota below the process count: the total rate is flat from %s processes on -- shared memory system (caches / DRAM channels)" % \
            next((k for k in sorted(sweep, key=int) if sweep[k]["value"] >= 0.9 * out["best"]), str(len(main)))
    else:
        plain = _cpu_run(equations, spaceDiscr, min(seconds, 5.0), cores, dims, "block", False)
        if plain:
            out["blockResCore_twin"] = {"value": sum(plain) / 1e6, "cores": len(plain), "what": "blockette.F90:755-852 (the figure of rounds 1-2)"}
    return out


T_START = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def rank_grid(n):
    """ranks as an rx x ry x rz grid, as cubic as possible (8 -> 2x2x2: three face peers + edge / corner peers per GPU)"""
    g = [1, 1, 1]
    d = 0
    while n > 1:
        f = 2 if n % 2 == 0 else n
        g[d % 3] *= f
        n //= f
        d += 1
    return tuple(g)


def weak_owner(e, rx, ry):
    """owner rank of global block g when every rank of an rx x ry x rz grid owns an e x e x e brick of blocks (blocks numbered with i
    fastest over the whole (e rx) x (e ry) x (e rz) brick); tests/mp_halo_worker.py runs the same function at 2 / 4 / 8 ranks"""
    Bi, Bj = e * rx, e * ry

    def owner(g):
        bi, bj, bk = g % Bi, (g // Bi) % Bj, g // (Bi * Bj)
        return (bi // e) + rx * ((bj // e) + ry * (bk // e))
    return owner


def strong_owner(nb, world):
    """owner rank of global block g when ONE mesh of nb equal blocks is farmed over `world` ranks (greedy bin-pack on the cell count,
    loadBalance.F90:409, on equal blocks: nb / world each, taken in index order so that a rank's blocks are neighbours)"""
    own = {g: (g * world) // nb for g in range(nb)}
    return lambda g: own[g]


def w_cycle(nlev):
    """cycleStrategy of an `nlev`w cycle (inputParamRoutines.F90:1127-1180 setEntriesWcycle)"""
    if nlev == 2:
        return [0, 1, 0, -1]
    return [0, 1] + w_cycle(nlev - 1) + w_cycle(nlev - 1) + [0, -1]


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU)."""
    import torch
    have = torch.cuda.device_count()
    if have < a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} GPU(s) visible on this node")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


class Job:
    """One workload resident on this rank's GPU."""

    def __init__(self, a, name, eng, rank, world, levels=1, keep_w=False):
        from adflow_amd.params import FlowParams
        from adflow_amd.synth import make_block, make_coarse_block
        from adflow_amd.topology import BrickTopology
        import numpy as np
        self.name, self.eng, self.rank, self.world, self.a = name, eng, rank, world, a
        wl = self.wl = WORKLOADS[name]
        self.prm = FlowParams(equations=wl["equations"], spaceDiscr=wl["spaceDiscr"], vis4=0.1 if wl["spaceDiscr"] == 2 else 0.0156)
        eng.set_options(self.prm)
        nb, dims = wl["nblocks"], wl["dims"]
        e = round(nb ** (1.0 / 3.0))                 # brick of e x e x e blocks
        assert e ** 3 == nb
        self.scaling = getattr(a, "scaling", "weak")
        if self.scaling == "strong":
            # BASELINE.json north_star / configs[3]: the SAME mesh (one e^3 brick, periodic) farmed over the GPUs, whole blocks to
            # ranks by a greedy bin-pack on the cell count (the criterion of loadBalance.F90:409; equal blocks here, so every rank
            # gets nb / N of them, neighbours in k first): total work fixed, the exchange grows with N
            rx = ry = rz = 1
            self.grid = (1, 1, 1)
            owner = strong_owner(nb, world)
        else:
            # weak scaling: every GPU owns an e^3 brick of blocks; the ranks form an rx x ry x rz grid of such bricks, periodic
            # in all three directions, so every evaluation is preceded by the 2-layer exchange of blocketteRes (whalo2,
            # blockette.F90:246): same-GPU copies + RCCL send/recv with up to 7 distinct peers at 8 ranks
            rx, ry, rz = rank_grid(world)
            self.grid = (rx, ry, rz)
            owner = weak_owner(e, rx, ry)
        self.bc = wl.get("bc")
        periodic = (False, False, False) if self.bc else (True, True, True)
        self.topo = [BrickTopology(e * rx, e * ry, e * rz, dims[0] >> l, dims[1] >> l, dims[2] >> l, owner=owner, periodic=periodic)
                     for l in range(levels)]
        lid = self.topo[0].local_ids()
        self.cells_local = 0
        self.wvec = []
        self.nsubfaces = self.nwallfaces = 0
        self._keep = []
        for g in self.topo[0].blocks_of(rank):
            if self.bc:
                # wall-bounded brick: the block sits at its place in the brick (wall distance measured from the brick's floor), only
                # the blocks on the floor are clustered towards it
                cg = self.topo[0].coords(g)
                blk = make_block(*dims, self.prm, seed=20260925 + g, stretch_k=3.0 if cg[2] == 0 else 1.0, wall_kmin=False)
                blk["d2Wall"] += float(cg[2])
            else:
                blk = make_block(*dims, self.prm, seed=20260925 + g, stretch_k=3.0 if wl["equations"] == 3 else 1.0)
            chain = [blk]
            for l in range(1, levels):
                chain.append(make_coarse_block(chain[-1], self.prm, seed=777 * l + g))   # also attaches the mg maps to the finer block
            faces, nvisc = [], 0
            if self.bc:
                from adflow_amd.synth import make_bocos, set_porosities
                spec = self.topo[0].boundary_spec(g, self.bc)
                if spec:
                    faces, nvisc = make_bocos(blk, self.prm, spec, seed=31 * g + 1)
                set_porosities(blk, faces)
            for l, b_ in enumerate(chain):
                eng.register(b_, nn=lid[g], level=l + 1)
            if faces:
                eng.bc_register(faces, nvisc, nn=lid[g], level=1)
                self._keep.append(faces)          # BCData arrays: the library keeps device copies, the host arrays stay valid anyway
                self.nsubfaces += len(faces)
                self.nwallfaces += sum((f["icEnd"] - f["icBeg"] - 1) * (f["jcEnd"] - f["jcBeg"] - 1) for f in faces[:nvisc])
            self.cells_local += blk.ncells
            if keep_w:   # PETSc vector order: block, k, j, i, variable fastest (NKSolvers.F90:1240-1253)
                wo = blk["w"][2:blk.il + 1, 2:blk.jl + 1, 2:blk.kl + 1, :]
                self.wvec.append(np.ascontiguousarray(wo.transpose(2, 1, 0, 3)).ravel())
            for b_ in chain:       # host copies are no longer needed by the timed loops
                for k in list(b_.a.keys()):
                    if k not in ("dw",):
                        del b_.a[k]
            if nb <= 16 or lid[g] % 49 == 0 or lid[g] == nb:
                log(f"{name}: block {lid[g]}/{nb} generated and uploaded")
        self.halo = "off"
        self.cp = []
        try:
            for l in range(levels):
                cp = self.topo[l].patterns(2 if l == 0 else 1, only_rank=rank)[rank]
                eng.comm_register(l + 1, 2 if l == 0 else 1, cp)
                self.cp.append(cp)
            cp = self.cp[0]
            log(f"comm pattern: {cp.ncopy} local copies, {int(cp.nsendCum[-1])} cells sent to {cp.sendProc.size} peer ranks")
            self.peers = int(cp.sendProc.size)
            eng.whalo2(1, 1, self.prm.nw)
            self.halo = "whalo2 every step: same-GPU copies" + (f" + RCCL send/recv over xGMI with {self.peers} peers" if world > 1 else "")
        except Exception as ex:  # the evaluation of independent shards is still a valid measurement
            self.halo = f"FAILED ({ex}); shards evaluated without exchange"
            log("halo exchange unavailable: " + str(ex))
        self.do_halo = not self.halo.startswith("FAILED")

    def step(self):
        if self.bc:
            # the reference's whole blocketteRes in ONE call (blockette.F90:199-283, default flags: storeWall = T): derived values of
            # the owned cells, turbulence + mean-flow boundary conditions of every subface, whalo2, core, viscSubface%tau / %q
            self.eng.blocketteRes(1, False, True, self.wl["equations"] == 3, halo=True, closures=True)
            return
        # blocketteRes with the reference's default flags: updateIntermed = F, flowRes = T, turbRes = T.  The exchange in front of
        # the core is whalo2 either way; inside ONE call (default) the library may run the tiles that read no halo cell while the
        # messages are in flight (api.hip block_res_split_enqueue; only when the pattern has messages, i.e. N > 1 or comm_self)
        if self.do_halo and not getattr(self.a, "separate_halo", False):
            self.eng.blocketteRes(1, False, True, self.wl["equations"] == 3, halo=True)
            return
        if self.do_halo:
            self.eng.whalo2(1, 1, self.prm.nw)
        self.eng.blocketteRes(1, False, True, self.wl["equations"] == 3)


def timed(eng, fn, steps, barrier, min_seconds=1.0, max_reps=2000, agree=None):
    """Time `steps` calls of fn, repeated until the region lasts >= min_seconds (the driver's busy sampler needs that);
    returns (seconds per step, repeats, event ms per step)."""
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier()
    probe = time.perf_counter() - t0
    if agree is not None:
        probe = agree(probe)      # every rank must repeat the same number of times: each step holds a halo exchange
    reps = int(min(max_reps, max(1, -(-min_seconds // max(probe, 1e-6)))))
    barrier()
    t0 = time.perf_counter()
    eng.event_record(0)
    for _ in range(reps):
        for _ in range(steps):
            fn()
    eng.event_record(1)
    barrier()
    dt = time.perf_counter() - t0
    return dt / (reps * steps), reps, eng.event_elapsed_ms(0, 1) / (reps * steps)


def phase_times(eng, fn, n=10, names=PHASES):
    """live HIP-event durations (ms) between the phase marks of blocketteRes, averaged over n evaluations"""
    base = 40
    eng.set_tuning("phase_events", base)
    acc = [0.0] * 6
    ok = [0] * 6
    for _ in range(n):
        fn()
        eng.sync()
        last = 0
        for m in range(1, 7):
            try:
                acc[m - 1] += eng.event_elapsed_ms(base + last, base + m)
                ok[m - 1] += 1
                last = m
            except Exception:
                pass        # mark not recorded by this configuration (e.g. no viscous part)
    eng.set_tuning("phase_events", 0)
    return {names[m]: acc[m] / ok[m] for m in range(6) if ok[m]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the other configurations reported under 'extra' (N=1)")
    ap.add_argument("--force-extras", action="store_true", help="run the extras although --tuning is given (A/B runs of a tuning key)")
    ap.add_argument("--no-mg", action="store_true", help="skip the config-2 multigrid cycle measurement")
    ap.add_argument("--only-extras", default="", help="comma list out of 4b,matvec,pc,config3,periodic,config2,config1,shard,drdw,small (drdw8, drdw_jst, pc_jst on request only): time only these extras (kernel traces)")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="repeat the timed K steps until the region lasts this long")
    ap.add_argument("--tuning", action="append", default=[], help="key=value knobs of adflow_gpu_set_tuning")
    ap.add_argument("--separate-halo", action="store_true",
                    help="whalo2 and the blocketteRes core as two calls (rounds 1-2) instead of one call with the exchange inside")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the driver's contract): every GPU owns its own brick of the workload's blocks; strong: the SAME "
                         "mesh (BASELINE north_star: the 8-block CRM mesh) split over the GPUs, 8 / N blocks each")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    log("torch imported")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world)

    import numpy as np
    from adflow_amd import capi
    from adflow_amd.engine import Engine
    from adflow_amd.params import DADI, RungeKutta, alternateResAveraging, noResAveraging

    eng = Engine(local_rank)
    tuning = dict(kv.split("=") for kv in a.tuning)
    for k_, v_ in tuning.items():
        eng.set_tuning(k_, int(v_))

    # RCCL communicator of the library (also at N=1: exercises the bootstrap)
    import ctypes
    idbuf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        raw = (ctypes.c_char * 128)()
        capi.check(eng.lib.adflow_gpu_comm_unique_id(raw), eng.lib)
        idbuf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
    if world > 1:
        idg = idbuf.cuda()
        dist.broadcast(idg, 0)
        idbuf = idg.cpu()
    capi.check(eng.lib.adflow_gpu_comm_init(rank, world, idbuf.numpy().tobytes()), eng.lib)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    def agree(x):
        # the same number on every rank (maximum over ranks)
        if world == 1:
            return x
        t_ = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)
        return float(t_.item())

    extras_on = (world == 1 and not a.no_extras and a.workload == DEFAULT_WORKLOAD and (not a.tuning or a.force_extras))
    job = Job(a, a.workload, eng, rank, world, keep_w=extras_on)
    wl, prm = job.wl, job.prm
    eng.set_async(True)
    for _ in range(a.warmup):
        job.step()
    sec_step, reps, ev_ms = timed(eng, job.step, a.steps, barrier, a.min_seconds, agree=agree)
    log(f"timed loop done: {sec_step * 1e3:.3f} ms/step ({reps} x {a.steps} steps)")
    if world > 1:
        t = torch.tensor([sec_step], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sec_step = float(t.item())
    if world > 1:
        tc = torch.tensor([float(job.cells_local)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tc, op=dist.ReduceOp.SUM)
        cells_total = int(tc.item())
    else:
        cells_total = job.cells_local
    value = cells_total / sec_step / 1e6
    # ---- what every rank exchanges per step (round-4 verdict, next 8 ii): peers, cells and bytes sent / received, same-GPU copies, and
    # what the RCCL communicator itself reports -- the first multi-GPU run checks itself
    comm_info = None
    if job.cp:
        import ctypes as _ct
        cp0 = job.cp[0]
        nvar = prm.nw + 1 + (2 if wl["equations"] >= 2 else 0)          # w(1:nw), p, rlv, rev of whalo2
        r_, n_, cnt_, ur_ = _ct.c_int(), _ct.c_int(), _ct.c_int(), _ct.c_int()
        capi.check(eng.lib.adflow_gpu_comm_info(_ct.byref(r_), _ct.byref(n_), _ct.byref(cnt_), _ct.byref(ur_)), eng.lib)
        mine = [rank, int(cp0.sendProc.size), int(cp0.recvProc.size), int(cp0.nsendCum[-1]), int(cp0.nrecvCum[-1]), int(cp0.ncopy),
                int(cnt_.value), int(ur_.value)]
        rows = [mine]
        if world > 1:
            tg = torch.tensor(mine, dtype=torch.int64, device="cuda")
            allr = [torch.zeros_like(tg) for _ in range(world)]
            dist.all_gather(allr, tg)
            rows = [[int(v) for v in t_.tolist()] for t_ in allr]
        comm_info = {"variables_per_halo_cell": nvar,
                     "per_rank": [{"rank": r[0], "send_peers": r[1], "recv_peers": r[2], "cells_sent": r[3], "cells_received": r[4],
                                   "bytes_sent_per_step": r[3] * nvar * 8, "bytes_received_per_step": r[4] * nvar * 8,
                                   "same_gpu_copies": r[5], "ncclCommCount": r[6], "ncclCommUserRank": r[7]} for r in rows],
                     "checks": {"every_rank_in_one_communicator_of_world_size": all(r[6] == world and r[7] == r[0] for r in rows) if world > 1 else None,
                                "cells_sent_equal_cells_received_over_all_ranks": sum(r[3] for r in rows) == sum(r[4] for r in rows),
                                "every_rank_has_peers": all(r[1] > 0 and r[2] > 0 for r in rows) if world > 1 else None,
                                "exchange_registered_on_every_rank": len(rows) == world}}
        # a scaling line must never come from ranks that did not talk to each other (round-5 verdict, next 8): N > 1 without one
        # communicator of N ranks, without peers, or with unbalanced traffic is an ERROR (ok = false, exit code 3)
        comm_info["ok"] = all(v is not False for v in comm_info["checks"].values())
    comm_ok = (world == 1) or (comm_info is not None and comm_info["ok"] and job.do_halo)

    # ---- per-kernel durations of one evaluation, live HIP events on the library's stream
    eng.set_async(False)
    # NS / RANS over the tile table: the fused gradient + viscous march runs in front of the inviscid march (api.hip enqueue_flow_fluxes)
    gf_on = wl["equations"] >= 2

    def phase_names(first):
        if gf_on:
            return PHASES_GF_FIRST if first else PHASES_GF
        return PHASES
    ph = phase_times(eng, job.step, names=phase_names(True))
    log("phases (ms): " + ", ".join(f"{k} {v:.3f}" for k, v in ph.items()))
    # every phase between the step's events is part of the evaluation (the front part = derived values, boundary conditions, whalo2)
    kern = {("closures + BCs + whalo2" if k == "closures+bc" else k): v for k, v in ph.items() if k != "(mark)"}
    dom = max(kern, key=kern.get)

    extra = {}
    only = set(x for x in a.only_extras.split(",") if x)

    def want(name):
        return not only or name in only
    if extras_on:
        try:
            if want("4b"):
                # ---- config 4b: matrix dissipation on the same blocks
                p4b = prm.replace(spaceDiscr=2, vis4=0.1)
                eng.set_options(p4b)
                eng.set_async(True)
                for _ in range(3):
                    job.step()
                s4b, r4b, e4b = timed(eng, job.step, a.steps, barrier, a.min_seconds)
                eng.set_async(False)
                ph4b = phase_times(eng, job.step, names=phase_names(True))
                ph4b.pop("(mark)", None)
                extra["crm_rans_sa_matrix_8x160x128x64"] = {
                    "value": job.cells_local / s4b / 1e6, "unit": "Mcells*residual-evals/s", "ms_per_step": s4b * 1e3,
                    "whole_eval_hbm_frac": 255.0 * job.cells_local / s4b / 1e9 / HBM_PEAK_GBS, "phase_ms": ph4b}
                log(f"4b matrix: {s4b * 1e3:.3f} ms/step")
                eng.set_options(prm)
            if want("matvec"):
                # ---- config 5: GMRES proxy, 30 matrix-free matvecs FormFunction_mf(w + h v_k) (NKSolvers.F90:437-461), vectors on the device
                n = sum(v.size for v in job.wvec)
                w0 = torch.from_numpy(np.concatenate(job.wvec)).cuda()
                job.wvec = []
                gen = torch.Generator(device="cuda").manual_seed(5)
                vk = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) - 0.5
                vk /= vk.norm()
                wk = w0 + 1e-7 * vk
                rv = torch.empty_like(w0)
                torch.cuda.synchronize()

                def matvec():
                    capi.check(eng.lib.adflow_gpu_nk_residual_dev(wk.data_ptr(), rv.data_ptr(), n), eng.lib)
                eng.set_async(True)
                for _ in range(3):
                    matvec()
                s5, r5, e5 = timed(eng, matvec, 30, barrier, a.min_seconds)
                eng.set_async(False)
                extra["config5_gmres_proxy"] = {
                    "value": job.cells_local / s5 / 1e6, "unit": "Mcells*residual-evals/s", "ms_per_matvec": s5 * 1e3,
                    "ms_per_30_matvecs": 30 * s5 * 1e3, "bytes_per_cell": 255.0 + 96.0,
                    "whole_eval_hbm_frac": (255.0 + 96.0) * job.cells_local / s5 / 1e9 / HBM_PEAK_GBS,
                    "what": "setW + blocketteRes(default flags: closures, BCs, whalo2, core) + setRVec at w + 1e-7 v, device vectors"}
                log(f"config 5 proxy: {s5 * 1e3:.3f} ms/matvec")
                del w0, vk, wk, rv
            if want("pc"):
                # restore the state (setW clipped / perturbed it by 1e-7) is not needed: the remaining extras re-time only
                # ---- the preconditioner matrix of NK / ANK: setupStateResidualMatrix(usePC = T, useAD = F), adjointUtils.F90:7-715 --
                # 7 colours x 6 states coloured finite differences, every one closures + boundary conditions + approximate residual +
                # the scatter of one column of all 7 stencil blocks; the blocks (7 x 36 doubles per cell) stay on the device
                eng.setupStateResidualMatrix(1, usePC=True)            # first call allocates the block storage
                barrier()
                t0 = time.perf_counter()
                eng.setupStateResidualMatrix(1, usePC=True)
                barrier()
                spc = time.perf_counter() - t0
                extra["pc_matrix_assembly"] = {
                    "ms": spc * 1e3, "residual_evaluations": 43, "ms_per_evaluation": spc * 1e3 / 43.0,
                    "what": "adflow_gpu_fd_jacobian(PC): 7 colours x 6 states + the reference evaluation, lumped Roe dissipation, "
                            "thin-layer viscous flux, first-order SA advection; blocks resident in HBM (2016 B per cell)"}
                log(f"PC matrix assembly: {spc * 1e3:.1f} ms ({spc * 1e3 / 43.0:.2f} ms per coloured evaluation)")
                # the same matrix by forward mode (useAD = T, adjointUtils.F90:227-409): 42 dual-number evaluations of the gather
                # kernels (kernels_ad.hip) -- exact derivatives, no step size; not tuned (the marching kernels have no dual form)
                eng.setupStateResidualMatrix(1, usePC=True, useAD=True)      # first call lays out the slab of dual arrays (13.5 GB here)
                barrier()
                t0 = time.perf_counter()
                eng.setupStateResidualMatrix(1, usePC=True, useAD=True)
                barrier()
                sad = time.perf_counter() - t0
                eng.set_tuning("ad_cache", 0)                                # hand the slab back before the other extras
                eng.set_tuning("ad_cache", 1)
                extra["pc_matrix_assembly_forward_ad"] = {
                    "ms": sad * 1e3, "forward_evaluations": 42, "ms_per_evaluation": sad * 1e3 / 42.0,
                    "what": "adflow_gpu_fd_jacobian(PC | USE_AD): seed = 1 on one state variable of one colour per pass; the marching "
                            "kernels of the approximate residual on dual numbers (k_pc_march: first-order Roe + thin-layer viscous flux, "
                            "k_sa_march), dual closures and boundary conditions, snapshots written by the marches; the dual copies of the "
                            "level's arrays (640 B per box cell, one slab kept between calls) are refreshed from the library's arrays "
                            "inside the call"}
                log(f"PC matrix assembly, forward AD: {sad * 1e3:.1f} ms")
            if "pc_jst" in only:
                # (only on request) the same two assemblies with the central scheme + scalar JST: lumped scalar dissipation with the
                # frozen sensor (inviscidDissFluxScalarApprox, fluxes.F90:3861-4342) + thin-layer viscous flux
                eng.set_options(prm.replace(spaceDiscr=1))
                res_j = {}
                for label, ad in (("finite_differences", False), ("forward_ad", True)):
                    eng.setupStateResidualMatrix(1, usePC=True, useAD=ad)
                    barrier()
                    t0 = time.perf_counter()
                    eng.setupStateResidualMatrix(1, usePC=True, useAD=ad)
                    barrier()
                    res_j[label + "_ms"] = (time.perf_counter() - t0) * 1e3
                eng.releaseWorkspace()
                eng.set_options(prm)
                extra["pc_matrix_assembly_scalar_jst"] = res_j
                log("PC matrix assembly, scalar JST: " + ", ".join(f"{k} {v:.1f}" for k, v in res_j.items()))
            if "drdw8" in only:
                # (only on request: 112 GB of stencil blocks) the exact dR/dw of the adjoint by forward mode on the whole 8-block mesh
                eng.setupStateResidualMatrix(1, usePC=False, useAD=True)
                barrier()
                t0 = time.perf_counter()
                eng.setupStateResidualMatrix(1, usePC=False, useAD=True)
                barrier()
                sdr8 = time.perf_counter() - t0
                extra["exact_drdw_forward_ad"] = {"ms": sdr8 * 1e3, "forward_evaluations": 210, "ms_per_evaluation": sdr8 * 1e3 / 210.0,
                                                  "what": "adflow_gpu_fd_jacobian(USE_AD) without PC on the whole mesh (33-point stencil blocks)"}
                log(f"exact dR/dw, forward mode, 8 blocks: {sdr8 * 1e3:.1f} ms")
                eng.releaseWorkspace()
            if want("config3"):
                # ---- config 3: one solver iteration = D-ADI x3 sub-iterations + SA DDADI x3 (test_functionals.py:136-160)
                eng.set_options(prm.replace(spaceDiscr=1, smoother=DADI, nSubiterations=3, nSubIterTurb=3, cfl=1.5, resAveraging=noResAveraging))
                eng.timeStep(1, False)
                eng.residual(1, 0)
                for _ in range(2):
                    eng.executeMGCycle([0])
                s3, r3, e3 = timed(eng, lambda: eng.executeMGCycle([0]), 5, barrier, a.min_seconds)
                b_res, b_dadi, b_upd = 255.0 + 32.0, 244.0, 224.0
                extra["config3_dadi_iteration"] = {
                    "iterations_per_s": 1.0 / s3, "ms_per_iteration": s3 * 1e3,
                    "what": "single grid on the same blocks with scalar JST: D-ADI x3 sub-iterations + SA DDADI x3",
                    "hbm_frac": 3 * (b_res + b_dadi + b_upd) * job.cells_local / s3 / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_bytes_formula": "3 x (B_res 287 + B_dadi 244 + B_upd 224) B per cell, SA solve not counted"}
                log(f"config 3 iteration: {s3 * 1e3:.3f} ms")
                eng.set_options(prm)
        except Exception as ex:
            extra["error_rans_extras"] = str(ex)
            log("RANS extras failed: " + str(ex))

    out = None
    if rank == 0:
        # HBM-side bytes per launch from the PMC passes (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc runs,
        # tools/pmc_traffic.py writes the file with the git hash it was measured at; the raw summaries sit beside it)
        traffic, traffic_src = None, None
        try:
            tf = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            ent = tf.get(a.workload)
            if ent and not a.tuning:
                traffic, traffic_src = ent, ent.get("source")
        except (OSError, ValueError, KeyError):
            pass
        alg_bytes = wl["bytes_per_cell"] * job.cells_local        # the figure builder and judge share
        eval_ms = sum(kern.values())
        dom_traffic = None
        if traffic:
            dom_traffic = traffic.get("kernels", {}).get(dom, {}).get("traffic_bytes_per_launch")
        # the evaluation's kernels overlap on the library's side streams: the roofline figure prices the TIMED evaluation (HIP events
        # around the K steps on the library's stream, halo copies included), kernels_ms lists them measured one after the other
        achieved = alg_bytes / (ev_ms * 1e-3) / 1e9
        cal = load_calibration()
        stream_gbs = float(cal.get("copy16u8_1gib_gbs", HBM_STREAM_GBS))
        out = {
            "metric": "Mcells*residual-evals/s", "value": value, "unit": "Mcells*residual-evals/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": sec_step * 1e3, "repeats": reps,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{a.workload}: {wl['nblocks']} blocks x {wl['dims'][0]}x{wl['dims'][1]}x{wl['dims'][2]} cells per GPU, "
                                   + wl["desc"] + (", one residual evaluation per step = the reference's whole blocketteRes (derived values, "
                                                   "turbulence + mean-flow BCs, whalo2, core with storeWall; default flags)" if job.bc else
                                                   ", one residual evaluation per step (whalo2 + blocketteRes core, default flags)"),
                       "boundary_subfaces": job.nsubfaces, "viscous_wall_faces": job.nwallfaces,
                       "halo_exchange": job.halo, "rank_grid": "x".join(map(str, job.grid)),
                       "partition": ("strong: one periodic brick of %d blocks, %d per rank (contiguous in k)" % (wl["nblocks"], wl["nblocks"] // world))
                       if a.scaling == "strong" else "weak: one brick of the workload's blocks per rank",
                       "cells_per_gpu": job.cells_local, "device": eng.device_name()},
            # the evaluation is several kernels (SA, inviscid, nodal gradients, viscous): `achieved` prices the WHOLE timed
            # evaluation against the 255 / 175 B per cell of SURVEY §8(d); dominant_kernel is the longest of them
            "comm": comm_info if comm_info is not None else ({"ok": False, "error": job.halo} if world > 1 else None),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (traffic or {}).get("traffic_bytes_per_eval"), "traffic_source": traffic_src,
                         "algorithmic_bytes_per_eval": alg_bytes,
                         "algorithmic_bytes_per_cell": {"core (SURVEY 8d)": wl["bytes_per_cell"],
                                                        "derived values (p, rlv, rev from w), not in `achieved`": wl.get("bytes_front", 0.0)},
                         # secondary: the same time priced with the 72 B per cell of the derived-values pass the wall-bounded step
                         # also runs (round-4 verdict, weak 4: `frac` is the contract's 255 B figure)
                         "frac_with_derived_values": (wl["bytes_per_cell"] + wl.get("bytes_front", 0.0)) * job.cells_local / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "frac_core_bytes_only": wl["bytes_per_cell"] * job.cells_local / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "eval_ms": ev_ms, "kernels_ms": kern, "kernels_ms_sum_serial": eval_ms,
                         "dominant_kernel": dom, "dominant_kernel_ms": kern[dom], "dominant_kernel_traffic": dom_traffic,
                         "dominant_kernel_share": kern[dom] / eval_ms,
                         # context, not the priced peak: the streaming-copy ceiling measured on this pool and the fraction of it the
                         # evaluation reaches for the bytes the counters say it actually moves
                         "streaming_copy_gbs": stream_gbs,
                         "streaming_copy_source": "profiles/calibration.json (tools/pmc_calib.bin bw2: 16 B per lane, 8 loads in flight per "
                                                  "thread, 1 GiB arrays)" if "copy16u8_1gib_gbs" in cal else "MI355X_MICROARCH.md (float4 copy)",
                         "infinity_cache_copy_gbs": cal.get("copy16u8_32mib_gbs"),
                         # the second bound: FP64 VALU issue (SURVEY section 8(d) asks for it beside GB/s)
                         "fp64_issue": fp64_issue_roofline(eng, ev_ms, kern, gf=gf_on, spaceDiscr=wl["spaceDiscr"]) if wl["equations"] == 3 else None,
                         "traffic_gbs": ((traffic or {}).get("traffic_bytes_per_eval") or 0.0) / (ev_ms * 1e-3) / 1e9 or None},
            "whole_eval": {"event_ms_per_step": ev_ms, "overlap_gain_ms": eval_ms - ev_ms,
                           "hbm_frac": alg_bytes / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "traffic_over_algorithmic": ((traffic or {}).get("traffic_bytes_per_eval") or 0.0) / alg_bytes or None},
        }

    # ---- the fully periodic brick of rounds 1-3 (no subfaces, step = whalo2 + core): continuity with BENCH_r01..r03
    if extras_on and want("periodic"):
        try:
            eng.release_all()
            eng.set_options(prm)
            jp = Job(a, PERIODIC_TWIN, eng, rank, world)
            eng.set_async(True)
            for _ in range(3):
                jp.step()
            sp_, rp_, ep_ = timed(eng, jp.step, a.steps, barrier, a.min_seconds)
            eng.set_async(False)
            php = phase_times(eng, jp.step, names=phase_names(True))
            php.pop("(mark)", None)
            extra[PERIODIC_TWIN] = {"value": jp.cells_local / sp_ / 1e6, "unit": "Mcells*residual-evals/s", "ms_per_step": sp_ * 1e3,
                                    "whole_eval_hbm_frac": 255.0 * jp.cells_local / sp_ / 1e9 / HBM_PEAK_GBS, "phase_ms": php,
                                    "what": "the headline workload of rounds 1-3: the same 8 blocks as a fully periodic brick without "
                                            "boundary subfaces, step = whalo2 + blocketteRes core (no derived values, no BCs)"}
            log(f"periodic twin: {sp_ * 1e3:.3f} ms/step")
            del jp
        except Exception as ex:
            extra["error_periodic_twin"] = str(ex)
            log("periodic twin failed: " + str(ex))
    # ---- config 2: Euler JST + 3-level W multigrid cycle (N=1 extras, or the workload itself when asked for)
    if extras_on and not a.no_mg and want("config2"):
        try:
            eng.release_all()
            job = None
            j2 = Job(a, "euler_jst_8x128", eng, rank, world, levels=3)
            eng.set_async(True)
            for _ in range(3):
                j2.step()
            s2, r2, e2 = timed(eng, j2.step, a.steps, barrier, a.min_seconds)
            eng.set_async(False)
            ph2 = phase_times(eng, j2.step)
            extra["euler_jst_8x128"] = {"value": j2.cells_local / s2 / 1e6, "unit": "Mcells*residual-evals/s", "ms_per_step": s2 * 1e3,
                                        "whole_eval_hbm_frac": 175.0 * j2.cells_local / s2 / 1e9 / HBM_PEAK_GBS, "phase_ms": ph2}
            log(f"config 2 Euler: {s2 * 1e3:.3f} ms/step")
            # pyADflow defaults (pyADflow.py:5697-5731): RK smoother, "alternate" residual averaging, cflCoarse 1.0, fcoll 0.8
            eng.set_options(j2.prm.replace(smoother=RungeKutta, resAveraging=alternateResAveraging))
            cyc = w_cycle(3)
            eng.timeStep(1, False)
            eng.residual(1, 0)
            # ---- BASELINE config 1's sweep rate: ONE RungeKuttaSmoother sweep on the fine level (smoothers.F90:4-382: five stages,
            # each update -> boundary conditions -> whalo -> residual; residual averaging on the alternate stages)
            for _ in range(2):
                eng.RungeKuttaSmoother(1)
            srk, rrk, erk = timed(eng, lambda: eng.RungeKuttaSmoother(1), 3, barrier, min(a.min_seconds, 0.5))
            extra["rk5_sweep_8x128"] = {"sweeps_per_s": 1.0 / srk, "ms_per_sweep": srk * 1e3, "value": j2.cells_local / srk / 1e6,
                                        "unit": "Mcells*RK5-sweeps/s",
                                        "what": "one RungeKuttaSmoother sweep (5 stages, alternate residual averaging, exchange between the "
                                                "stages) on 8 x 128^3 Euler JST",
                                        "algorithmic_bytes_per_cell": 5 * (175.0 + 200.0),
                                        "hbm_frac": 5 * (175.0 + 200.0) * j2.cells_local / srk / 1e9 / HBM_PEAK_GBS}
            log(f"RK5 sweep: {srk * 1e3:.3f} ms")
            eng.timeStep(1, False)
            eng.residual(1, 0)
            for _ in range(2):
                eng.executeMGCycle(cyc)
            smg, rmg, emg = timed(eng, lambda: eng.executeMGCycle(cyc), 3, barrier, a.min_seconds)
            # algorithmic bytes of the cycle: per level visit nStages x (B_res + B_upd); level l has N / 8^l cells
            b_res, b_upd, b_ts, b_tr = 175.0, 200.0, 32.0, 8.0 * (2 * 5 + 2)
            lvl, per_cell = 0, 0.0
            for c_ in cyc:
                if c_ == 0:
                    per_cell += 5 * (b_res + b_upd) / 8 ** lvl
                elif c_ == 1:
                    per_cell += (b_res + b_ts + b_tr) / 8 ** lvl + b_res / 8 ** (lvl + 1)
                    lvl += 1
                else:
                    lvl -= 1
                    per_cell += b_tr / 8 ** lvl
            per_cell += b_res + b_ts
            extra["mg_3w_cycle"] = {"cycles_per_s": 1.0 / smg, "ms_per_cycle": smg * 1e3, "cycle": "3w: " + " ".join(map(str, cyc)),
                                    "what": "RK5 + alternate residual averaging on every visit, cflCoarse 1.0, vis2Coarse 0.5, fcoll 0.8",
                                    "fine_cells_per_gpu": j2.cells_local, "algorithmic_bytes_per_fine_cell": per_cell,
                                    "hbm_frac": per_cell * j2.cells_local / smg / 1e9 / HBM_PEAK_GBS}
            log(f"3w MG cycle: {smg * 1e3:.3f} ms")
        except Exception as ex:
            extra["error_config2"] = str(ex)
            log("config 2 extras failed: " + str(ex))
    # ---- BASELINE config 1 at roofline size: single-block Euler JST, step = whalo2 + blocketteRes core (time step + residual)
    if extras_on and want("config1"):
        for name1 in ("euler_jst_1x512x256x2", "euler_jst_1x192"):
            try:
                eng.release_all()
                job = None
                j1 = Job(a, name1, eng, rank, world)
                eng.set_async(True)
                for _ in range(3):
                    j1.step()
                s1, r1, e1 = timed(eng, j1.step, a.steps, barrier, min(a.min_seconds, 0.5))
                eng.set_async(False)
                extra[name1] = {"value": j1.cells_local / s1 / 1e6, "unit": "Mcells*residual-evals/s", "ms_per_step": s1 * 1e3,
                                "cells_per_gpu": j1.cells_local,
                                "whole_eval_hbm_frac": 175.0 * j1.cells_local / s1 / 1e9 / HBM_PEAK_GBS,
                                "what": WORKLOADS[name1]["desc"] + " (periodic with itself); step = whalo2 + blocketteRes core"}
                log(f"{name1}: {s1 * 1e3:.3f} ms/step")
                del j1
            except Exception as ex:
                extra["error_" + name1] = str(ex)
                log(name1 + " failed: " + str(ex))
    # ---- de-risking N > 1 at N = 1 (round-3 verdict, next 5 i): the N = 8 strong-scaling shard, its exchange through RCCL to the own rank
    if extras_on and want("shard"):
        try:
            eng.release_all()
            eng.set_options(prm)
            js = Job(a, "crm_strong_shard_1x160x128x64", eng, rank, world)
            res = {}
            for label, cs, sp in (("same_gpu_copies", 0, 1), ("rccl_self_no_split", 1, 0), ("rccl_self_split", 1, 1)):
                eng.set_tuning("comm_self", cs)
                eng.set_tuning("split_eval", sp)
                eng.set_async(True)
                for _ in range(3):
                    js.step()
                ss_, _, es_ = timed(eng, js.step, a.steps, barrier, min(a.min_seconds, 0.5))
                eng.set_async(False)
                res[label] = {"ms_per_step": ss_ * 1e3}
            eng.set_tuning("comm_self", 0)
            eng.set_tuning("split_eval", 1)
            halo_cells = int(js.cp[0].ncopy)
            extra["strong_shard_one_block"] = {
                "cells": js.cells_local, "halo_cells_exchanged": halo_cells, "bytes_received_per_eval": halo_cells * 9 * 8, **res,
                "exchange_through_rccl_ms": res["rccl_self_no_split"]["ms_per_step"] - res["same_gpu_copies"]["ms_per_step"],
                "split_gain_ms": res["rccl_self_no_split"]["ms_per_step"] - res["rccl_self_split"]["ms_per_step"],
                "what": "one 160x128x64 periodic block = what each GPU holds at N = 8 of --scaling strong; rccl_self: every interface "
                        "through k_halo_pack -> ncclSend / ncclRecv to the own rank -> k_halo_unpack (the code path of N > 1, executed on "
                        "one GPU: the transfer is a device copy, not xGMI); split: interior tiles between departure and arrival"}
            log("N = 8 shard (one block): " + ", ".join(f"{k} {v['ms_per_step']:.3f} ms" for k, v in res.items()))
            # ---- where the EXACT linearisation stands (round-5 verdict, next 7): dR/dw of the adjoint by forward mode on this one
            # block -- 35 colours x 6 states dual evaluations of the second-order Roe + full viscous + SA residual (round 6: k_visc_gf,
            # k_roe_march and k_sa_march compiled for dual numbers), 33-point stencil blocks (9.5 KB per cell) resident in HBM
            if want("drdw"):
                try:
                    eng.setupStateResidualMatrix(1, usePC=False, useAD=True)      # first call: block storage + the slab of dual arrays
                    barrier()
                    t0 = time.perf_counter()
                    eng.setupStateResidualMatrix(1, usePC=False, useAD=True)
                    barrier()
                    sdr = time.perf_counter() - t0
                    ns_, st_ = eng.jacobianInfo()
                    nev = 35 * ns_
                    extra["exact_drdw_forward_ad_1x160x128x64"] = {
                        "ms": sdr * 1e3, "forward_evaluations": nev, "ms_per_evaluation": sdr * 1e3 / nev, "cells": js.cells_local,
                        "ns_per_cell_and_evaluation": sdr * 1e9 / nev / js.cells_local,
                        "stencil_blocks": int(st_.shape[0]),
                        "what": "adflow_gpu_fd_jacobian(USE_AD) without PC: the adjoint's dR/dw (adjointUtils.F90:227-409), one 160x128x64 "
                                "block; for comparison the preconditioner matrix above costs pc_matrix_assembly_forward_ad.ms / 42 per "
                                "evaluation of 8 such blocks"}
                    log(f"exact dR/dw, forward mode, one block: {sdr * 1e3:.1f} ms ({sdr * 1e3 / nev:.2f} ms per evaluation)")
                    eng.releaseWorkspace()
                    if "drdw_jst" in only:
                        # the same with the central scheme + scalar JST dissipation (on request: --only-extras drdw_jst)
                        eng.set_options(js.prm.replace(spaceDiscr=1))
                        eng.setupStateResidualMatrix(1, usePC=False, useAD=True)
                        barrier()
                        t0 = time.perf_counter()
                        eng.setupStateResidualMatrix(1, usePC=False, useAD=True)
                        barrier()
                        sdj = time.perf_counter() - t0
                        extra["exact_drdw_forward_ad_jst_1x160x128x64"] = {"ms": sdj * 1e3, "forward_evaluations": nev,
                                                                           "ms_per_evaluation": sdj * 1e3 / nev}
                        log(f"exact dR/dw, forward mode, scalar JST, one block: {sdj * 1e3:.1f} ms")
                        eng.releaseWorkspace()
                        eng.set_options(js.prm)
                except Exception as ex:
                    extra["error_exact_drdw"] = str(ex)
                    log("exact dR/dw extra failed: " + str(ex))
            del js
        except Exception as ex:
            extra["error_strong_shard"] = str(ex)
            log("strong-shard extra failed: " + str(ex))
    # ---- small blocks: about the cell count of the headline in 343 blocks of 32^3 (a production multiblock mesh per GPU)
    if extras_on and want("small"):
        try:
            eng.release_all()
            j3 = Job(a, "rans_sa_upwind_343x32", eng, rank, world)
            eng.set_async(True)
            for _ in range(3):
                j3.step()
            s3b, r3b, e3b = timed(eng, j3.step, a.steps, barrier, a.min_seconds)
            eng.set_async(False)
            extra["rans_sa_upwind_343x32"] = {"value": j3.cells_local / s3b / 1e6, "unit": "Mcells*residual-evals/s", "ms_per_step": s3b * 1e3,
                                              "cells_per_gpu": j3.cells_local,
                                              "whole_eval_hbm_frac": 255.0 * j3.cells_local / s3b / 1e9 / HBM_PEAK_GBS,
                                              "what": "the headline evaluation on 343 blocks of 32^3 cells (7 x 7 x 7 periodic brick)"}
            log(f"343 x 32^3 blocks: {s3b * 1e3:.3f} ms/step")
            del j3
        except Exception as ex:
            extra["error_small_blocks"] = str(ex)
            log("small-block extra failed: " + str(ex))
    eng.close()
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        out["extra"] = extra or None
        if world == 1 and not a.no_cpu_baseline:
            log("cpu baseline (reference Fortran on the host cores) ...")
            out["cpu_baseline"] = cpu_baseline(wl["equations"], wl["spaceDiscr"])
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
    if not comm_ok:
        log(f"ERROR: --gpus {world}: the ranks did not exchange halos as one communicator (comm.ok = false): " + json.dumps(comm_info))
        sys.exit(3)


if __name__ == "__main__":
    main()
