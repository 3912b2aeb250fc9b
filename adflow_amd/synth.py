"""Synthetic structured blocks for the residual/smoother hot path.

The reference's regression meshes (CGNS) are downloaded at test time and are
absent here, so parity and benchmarks run on analytic
curvilinear blocks.  Arrays carry exactly the bounds the
reference allocates in Fortran (column-major) order:

    w,dw      (0:ib,0:jb,0:kb,1:nw)     p,gamma,rlv,rev,vol,iblank (0:ib,0:jb,0:kb)
    x         (0:ie,0:je,0:ke,3)        sI (0:ie,1:je,1:ke,3)  sJ (1:ie,0:je,1:ke,3)
    sK        (1:ie,1:je,0:ke,3)        porI (1:il,2:jl,2:kl)  porJ (2:il,1:jl,2:kl)
    porK      (2:il,2:jl,1:kl)          d2Wall (2:il,2:jl,2:kl)

Face metrics and volumes follow the reference's formulas
(`src/adjoint/adjointExtra.F90:5-178` volume_block, `:179-300` metric_block),
written here as whole-array numpy expressions.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np

from .params import FlowParams, normalFlux, boundFlux, noFlux


@dataclass
class Block:
    nx: int
    ny: int
    nz: int
    nw: int
    a: Dict[str, np.ndarray] = field(default_factory=dict)
    rotRate: Optional[tuple] = None      # cgnsDoms%rotRate of a moving block (blockIsMoving), None at rest
    rightHanded: bool = True             # blockType%rightHanded: (i, j, k) right-handed; False: metric_block uses fact = -half
    nodeParams: Optional[tuple] = None   # (xi, eta, zeta) of the nodes 0..ie / je / ke in the analytic map (make_nodes)
    coarsened: tuple = ("regular", "regular", "regular")   # blockType%iCoarsened / jCoarsened / kCoarsened (block.F90:230-233)
    frame: Optional[dict] = None         # LatticeTopology.frame: the block's place in the one analytic map of a multi-block mesh
    nodeMap: Optional[tuple] = None      # coarse block: per direction the FINE node of every coarse node 1..il (imap of createCoarseBlocks)

    # index helpers (reference naming)
    @property
    def il(self): return self.nx + 1
    @property
    def jl(self): return self.ny + 1
    @property
    def kl(self): return self.nz + 1
    @property
    def ie(self): return self.nx + 2
    @property
    def je(self): return self.ny + 2
    @property
    def ke(self): return self.nz + 2
    @property
    def ib(self): return self.nx + 3
    @property
    def jb(self): return self.ny + 3
    @property
    def kb(self): return self.nz + 3
    @property
    def ncells(self): return self.nx * self.ny * self.nz

    def __getitem__(self, k):
        return self.a[k]

    def __setitem__(self, k, v):
        self.a[k] = v

    def owned(self, name):
        """View of the owned cells (2:il,2:jl,2:kl[,:]) of a (0:ib,..) array."""
        return self.a[name][2:self.il + 1, 2:self.jl + 1, 2:self.kl + 1]

    def copy(self) -> "Block":
        return Block(self.nx, self.ny, self.nz, self.nw, {k: v.copy(order="F") for k, v in self.a.items()}, self.rotRate, self.rightHanded,
                     self.nodeParams, self.coarsened, self.frame, self.nodeMap)


def add_grid_velocities(blk: "Block", prm, rotRate=(0.05, -0.03, 0.12), rotCenter=(0.3, -0.2, 0.1)):
    """Make `blk` a block of a steadily rotating frame: sFaceI/J/K = (timeRef * rotRate x (face centre - rotCenter)) . S
    (what gridVelocitiesFineLevel + normalVelocitiesAllLevels leave in the block, solverUtils.F90:672-1190, 2063-2260)
    and blockIsMoving with cgnsDoms%rotRate for the rotational source term."""
    x = blk["x"]
    om = prm.timeRef * np.asarray(rotRate, float)
    c0 = np.asarray(rotCenter, float)

    def face(xs, S):
        v = np.cross(np.broadcast_to(om, xs.shape), xs - c0)
        return np.asfortranarray((v * S).sum(axis=-1))
    # face centre = mean of the four nodes of the face; node (i,j,k) of x(0:ie,0:je,0:ke) is x[i,j,k]
    xi = 0.25 * (x[:, :-1, :-1] + x[:, 1:, :-1] + x[:, :-1, 1:] + x[:, 1:, 1:])     # (0:ie, 1:je, 1:ke)
    xj = 0.25 * (x[:-1, :, :-1] + x[1:, :, :-1] + x[:-1, :, 1:] + x[1:, :, 1:])     # (1:ie, 0:je, 1:ke)
    xk = 0.25 * (x[:-1, :-1, :] + x[1:, :-1, :] + x[:-1, 1:, :] + x[1:, 1:, :])     # (1:ie, 1:je, 0:ke)
    blk["sFaceI"] = face(xi, blk["sI"])
    blk["sFaceJ"] = face(xj, blk["sJ"])
    blk["sFaceK"] = face(xk, blk["sK"])
    blk.rotRate = tuple(float(r) for r in rotRate)
    return blk


def F(shape, dtype=np.float64):
    return np.zeros(shape, dtype=dtype, order="F")


# ----------------------------------------------------------------------------
# geometry
# ----------------------------------------------------------------------------
def node_params(nx, ny, nz, stretch_k=1.0):
    """Parameters (xi, eta, zeta) of the nodes 0..ie / je / ke of a uniformly divided block (node 1 -> 0, node il -> 1)."""
    xi = (np.arange(nx + 3) - 1.0) / nx
    et = (np.arange(ny + 3) - 1.0) / ny
    ze = (np.arange(nz + 3) - 1.0) / nz
    if stretch_k != 1.0:
        # geometric-like clustering towards the k=kmin plane (wall)
        s = stretch_k
        ze = np.sign(ze) * (np.expm1(s * np.abs(ze)) / math.expm1(s))
    return xi, et, ze


def make_nodes(nx, ny, nz, lengths=(1.0, 1.0, 1.0), amp=0.02, stretch_k=1.0, origin=(0.0, 0.0, 0.0), params=None, frame=None):
    """Nodes x(0:ie,0:je,0:ke,3): smooth non-orthogonal right-handed map of the
    unit cube (node 1 -> 0, node il -> 1; halo nodes 0 and ie continue the map).
    params: the nodes' (xi, eta, zeta) instead of the uniform division (irregularly coarsened multigrid levels).
    frame (LatticeTopology.frame): the block is one of several that share ONE map; `params` are then the nodes' offsets from
    node 1 along the block's own index directions in lattice units, the lattice position of a node is o + T offsets and its
    (xi, eta, zeta) that position times `scale` -- blocks of any orientation and size get identical coordinates on shared nodes."""
    ie, je, ke = nx + 2, ny + 2, nz + 2
    xi, et, ze = params if params is not None else node_params(nx, ny, nz, stretch_k)
    assert len(xi) == ie + 1 and len(et) == je + 1 and len(ze) == ke + 1
    X, E, Z = np.meshgrid(xi, et, ze, indexing="ij")
    if frame is not None:
        T, o, sc = frame["T"], frame["o"], frame["scale"]
        P = [(o[a] + T[a, 0] * X + T[a, 1] * E + T[a, 2] * Z) * sc[a] for a in range(3)]
        X, E, Z = P
        amp = frame.get("amp", amp)
        sz = frame.get("stretch_z", 1.0)
        if sz != 1.0:
            Z = np.sign(Z) * (np.expm1(sz * np.abs(Z)) / math.expm1(sz))
    tp = 2.0 * math.pi
    x = F((ie + 1, je + 1, ke + 1, 3))
    x[..., 0] = origin[0] + lengths[0] * (X + amp * np.sin(tp * E) * np.sin(tp * Z))
    x[..., 1] = origin[1] + lengths[1] * (E + amp * np.sin(tp * X) * np.sin(tp * Z + 0.3))
    x[..., 2] = origin[2] + lengths[2] * (Z + amp * np.sin(tp * X + 0.7) * np.sin(tp * E))
    return x


def _cross(a, b):
    c = np.empty_like(a)
    c[..., 0] = a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1]
    c[..., 1] = a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2]
    c[..., 2] = a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]
    return c


def face_metrics(x):
    """sI(0:ie,1:je,1:ke,3), sJ(1:ie,0:je,1:ke,3), sK(1:ie,1:je,0:ke,3) — half the
    cross product of the face diagonals (metric_block, adjointExtra.F90:212-300)."""
    # i-faces: v1 = x(i,j,n) - x(i,m,k), v2 = x(i,j,k) - x(i,m,n), m=j-1, n=k-1
    v1 = x[:, 1:, :-1] - x[:, :-1, 1:]
    v2 = x[:, 1:, 1:] - x[:, :-1, :-1]
    sI = np.asfortranarray(0.5 * _cross(v1, v2))
    # j-faces: v1 = x(i,j,n) - x(l,j,k), v2 = x(l,j,n) - x(i,j,k), l=i-1
    v1 = x[1:, :, :-1] - x[:-1, :, 1:]
    v2 = x[:-1, :, :-1] - x[1:, :, 1:]
    sJ = np.asfortranarray(0.5 * _cross(v1, v2))
    # k-faces: v1 = x(i,j,k) - x(l,m,k), v2 = x(l,j,k) - x(i,m,k)
    v1 = x[1:, 1:, :] - x[:-1, :-1, :]
    v2 = x[:-1, 1:, :] - x[1:, :-1, :]
    sK = np.asfortranarray(0.5 * _cross(v1, v2))
    return sI, sJ, sK


def cell_volumes(x):
    """vol(0:ib,0:jb,0:kb) — six-pyramid split about the cell centroid
    (volume_block, adjointExtra.F90:40-110); 2nd-halo entries stay zero."""
    ie, je, ke = x.shape[0] - 1, x.shape[1] - 1, x.shape[2] - 1
    # corner views for cells 1..ie: (i or l=i-1, j or m=j-1, k or n=k-1)
    c = {}
    for a, sa in (("i", slice(1, None)), ("l", slice(0, -1))):
        for b, sb in (("j", slice(1, None)), ("m", slice(0, -1))):
            for d, sd in (("k", slice(1, None)), ("n", slice(0, -1))):
                c[a + b + d] = x[sa, sb, sd]
    cg = 0.125 * sum(c.values())

    def volpym(a, b, cc, d):
        q = 0.25 * (a + b + cc + d)
        ac = a - cc
        bd = b - d
        return ((cg[..., 0] - q[..., 0]) * (ac[..., 1] * bd[..., 2] - ac[..., 2] * bd[..., 1])
                + (cg[..., 1] - q[..., 1]) * (ac[..., 2] * bd[..., 0] - ac[..., 0] * bd[..., 2])
                + (cg[..., 2] - q[..., 2]) * (ac[..., 0] * bd[..., 1] - ac[..., 1] * bd[..., 0]))

    vp = (volpym(c["ijk"], c["ijn"], c["imn"], c["imk"])
          + volpym(c["ljk"], c["lmk"], c["lmn"], c["ljn"])
          + volpym(c["ijk"], c["ljk"], c["ljn"], c["ijn"])
          + volpym(c["imk"], c["imn"], c["lmn"], c["lmk"])
          + volpym(c["ijk"], c["imk"], c["lmk"], c["ljk"])
          + volpym(c["ijn"], c["ljn"], c["lmn"], c["imn"]))
    vol = F((ie + 2, je + 2, ke + 2))
    vol[1:ie + 1, 1:je + 1, 1:ke + 1] = np.abs(vp / 6.0)
    return vol


# ----------------------------------------------------------------------------
# state
# ----------------------------------------------------------------------------
def sutherland(prm: FlowParams, rho, p):
    """computeLamViscosity, flowUtils.F90:1201-1300 (no k correction for SA)."""
    muSuth = prm.muSuthDim / prm.muRef
    TSuth = prm.TSuthDim / prm.TRef
    SSuth = prm.SSuthDim / prm.TRef
    T = p / (prm.RGas * rho)
    return muSuth * ((TSuth + SSuth) / (T + SSuth)) * (T / TSuth) ** 1.5


def sa_eddy_viscosity(prm: FlowParams, rho, nut, rlv):
    """saEddyViscosity, turbUtils.F90:657-720: rev = rho*nuTilde*fv1."""
    cv13 = prm.SAcv1 ** 3
    chi = rho * nut / rlv
    chi3 = chi ** 3
    return rho * nut * chi3 / (chi3 + cv13)


def make_block(nx, ny, nz, prm: FlowParams, seed=20260925, lengths=(1.0, 1.0, 1.0), amp=0.02,
               stretch_k=1.0, wall_kmin=None, noise=0.02, wave=0.05, origin=(0.0, 0.0, 0.0), holes=0.0,
               noflux_jmax=False, moving=False, left_handed=False, params=None, frame=None) -> Block:
    """Analytic curvilinear block + perturbed free-stream state.
    moving: a block of a steadily rotating frame (add_grid_velocities).
    params: (xi, eta, zeta) of the nodes (make_nodes) instead of the uniform division; frame: see make_nodes."""
    rng = np.random.default_rng(seed)
    nw = prm.nw
    b = Block(nx, ny, nz, nw)
    ib, jb, kb = b.ib, b.jb, b.kb
    b.nodeParams = params if params is not None else node_params(nx, ny, nz, stretch_k)
    b.frame = frame
    x = make_nodes(nx, ny, nz, lengths, amp, stretch_k, origin, b.nodeParams, frame)
    sI, sJ, sK = face_metrics(x)
    if frame is not None and np.linalg.det(frame["T"]) < 0:
        assert not left_handed
        sI, sJ, sK = np.asfortranarray(-sI), np.asfortranarray(-sJ), np.asfortranarray(-sK)
        b.rightHanded = False
    if left_handed:
        # mirror image of the block: the index system becomes left-handed and the reference's metric_block takes fact = -half so
        # that the normals keep pointing towards increasing indices (adjointExtra.F90:205-211)
        x[..., 0] = -x[..., 0]
        sI, sJ, sK = face_metrics(x)
        sI, sJ, sK = np.asfortranarray(-sI), np.asfortranarray(-sJ), np.asfortranarray(-sK)
        b.rightHanded = False
    vol = cell_volumes(x)
    b["x"], b["sI"], b["sJ"], b["sK"], b["vol"] = x, sI, sJ, sK, vol
    b["volRef"] = vol.copy(order="F")

    # cell-centre parametric coordinates for all cells 0..ib
    ci = (np.arange(ib + 1) - 1.5) / nx
    cj = (np.arange(jb + 1) - 1.5) / ny
    ck = (np.arange(kb + 1) - 1.5) / nz
    X, E, Z = np.meshgrid(ci, cj, ck, indexing="ij")
    tp = 2.0 * math.pi
    winf = prm.wInf()
    gam = prm.gammaConstant

    def pert(phase):
        return 1.0 + wave * np.sin(tp * (3 * X + 2 * E + Z) + phase) + noise * rng.uniform(-1.0, 1.0, X.shape)

    w = F((ib + 1, jb + 1, kb + 1, nw))
    w[..., 0] = winf[0] * pert(0.0)
    w[..., 1] = winf[1] * pert(0.4)
    w[..., 2] = winf[1] * 0.1 * (pert(1.1) - 0.95) + winf[2] * pert(0.9)
    w[..., 3] = winf[1] * 0.1 * (pert(2.3) - 1.0)
    pr = prm.pInf * pert(1.7)
    v2 = w[..., 1] ** 2 + w[..., 2] ** 2 + w[..., 3] ** 2
    w[..., 4] = pr / (gam - 1.0) + 0.5 * w[..., 0] * v2
    b["w"] = w
    b["p"] = np.asfortranarray(pr)
    b["gamma"] = np.full((ib + 1, jb + 1, kb + 1), gam, order="F")
    b["iblank"] = np.ones((ib + 1, jb + 1, kb + 1), dtype=np.int32, order="F")

    porI = np.full((nx + 1, ny, nz), normalFlux, dtype=np.int8, order="F")
    porJ = np.full((nx, ny + 1, nz), normalFlux, dtype=np.int8, order="F")
    porK = np.full((nx, ny, nz + 1), normalFlux, dtype=np.int8, order="F")
    if wall_kmin is None:
        wall_kmin = prm.viscous
    if wall_kmin:
        porK[:, :, 0] = boundFlux  # solid wall on the k = kmin face
    if noflux_jmax:
        porJ[:, ny, :] = noFlux     # conservative non-matching boundary on the j = jmax face
    b["porI"], b["porJ"], b["porK"] = porI, porJ, porK
    if holes > 0.0:
        # overset-style blanking: iblank 0 (hole) and -1 (fringe) cells, halos included
        r = rng.uniform(0.0, 1.0, b["iblank"].shape)
        b["iblank"][r < holes] = 0
        b["iblank"][(r >= holes) & (r < 1.5 * holes)] = -1

    if prm.viscous:
        b["rlv"] = np.asfortranarray(sutherland(prm, w[..., 0], pr))
    else:
        b["rlv"] = F((ib + 1, jb + 1, kb + 1))
    b["rev"] = F((ib + 1, jb + 1, kb + 1))
    if nw > 5:
        nu_inf = prm.muInf / prm.rhoInf
        w[..., 5] = 3.0 * nu_inf * (1.0 + 0.5 * rng.uniform(0.0, 1.0, X.shape)) * (1.0 + 40.0 * np.clip(Z, 0, 1) * np.exp(-4 * np.clip(Z, 0, 1)))
        b["rev"] = np.asfortranarray(sa_eddy_viscosity(prm, w[..., 0], w[..., 5], b["rlv"]))
    # wall distance: distance from the cell centre to the k=kmin plane (>= 1e-6)
    xc = 0.125 * (x[1:, 1:, 1:] + x[:-1, 1:, 1:] + x[1:, :-1, 1:] + x[:-1, :-1, 1:]
                  + x[1:, 1:, :-1] + x[:-1, 1:, :-1] + x[1:, :-1, :-1] + x[:-1, :-1, :-1])
    d = xc[1:nx + 1, 1:ny + 1, 1:nz + 1, 2] - origin[2]
    b["d2Wall"] = np.asfortranarray(np.maximum(d, 1e-6))
    if moving:
        add_grid_velocities(b, prm)
    return b


# ----------------------------------------------------------------------------
# boundary subfaces: BCType / BCFaceID / BCData of a block whose six faces are physical
# boundaries (one subface per face, cell range including the first halo ring as the
# reference's preprocessing builds it).  Viscous walls come first (nViscBocos).
# ----------------------------------------------------------------------------
def make_bocos(blk: Block, prm: FlowParams, spec: dict, seed=7, split=(), split_at=None):
    """spec: {faceID (1..6 = iMin,iMax,jMin,jMax,kMin,kMax): BCType}.  Returns (faces, nViscBocos).
    split: {faceID: BCType of the second half}: the face is cut in two subfaces along its first index (the lower
    half keeps spec's kind), as block faces that are only partly a wall / partly farfield are in real meshes.
    split_at: {faceID: last cell of the lower subface} instead of the middle of the face (coarse multigrid levels: the cut
    follows the fine level's)."""
    rng = np.random.default_rng(seed)
    ie, je, ke = blk.ie, blk.je, blk.ke
    sI, sJ, sK = blk["sI"], blk["sJ"], blk["sK"]
    winf = prm.wInf()
    faces = []
    order = sorted(spec.items(), key=lambda kv: (0 if kv[1] in (-3, -4) else 1, kv[0]))
    for fid, typ in order:
        if fid in (1, 2):
            n = sI[1 if fid == 1 else blk.il, :, :, :]            # (je, ke, 3)
            rngs = (1, je, 1, ke)
        elif fid in (3, 4):
            n = sJ[:, 1 if fid == 3 else blk.jl, :, :]            # (ie, ke, 3)
            rngs = (1, ie, 1, ke)
        else:
            n = sK[:, :, 1 if fid == 5 else blk.kl, :]            # (ie, je, 3)
            rngs = (1, ie, 1, je)
        sgn = -1.0 if fid in (1, 3, 5) else 1.0                    # outward
        mag = np.sqrt((n ** 2).sum(axis=-1, keepdims=True))
        norm = np.asfortranarray(sgn * n / mag)
        shp = norm.shape[:2]
        f = dict(bcType=int(typ), faceID=int(fid), icBeg=rngs[0], icEnd=rngs[1], jcBeg=rngs[2], jcEnd=rngs[3], norm=norm)
        if typ in (-5, -6):
            f["rface"] = np.asfortranarray(0.01 * rng.uniform(-1, 1, shp))
        if typ in (-3, -4):
            f["uSlip"] = np.asfortranarray(0.02 * rng.uniform(-1, 1, shp + (3,)))
        if typ == -4:
            tinf = prm.pInf / (prm.RGas * prm.rhoInf)
            f["TNS_Wall"] = np.asfortranarray(tinf * (1.0 + 0.05 * rng.uniform(-1, 1, shp)))
        if typ in (-10, -12):       # subsonic outflow / outflow bleed: static pressure
            f["ps"] = np.asfortranarray(prm.pInf * (1.0 + 0.03 * rng.uniform(-1, 1, shp)))
        if typ == -8:               # subsonic inflow: total conditions on min faces, mass flow on max faces
            gam, R = prm.gammaConstant, prm.RGas
            tinf = prm.pInf / (R * prm.rhoInf)
            m2 = prm.Mach ** 2
            if fid in (1, 3, 5):
                f["subsonicInletTreatment"] = 1
                tt = tinf * (1.0 + 0.5 * (gam - 1.0) * m2) * (1.0 + 0.02 * rng.uniform(-1, 1, shp))
                pt = prm.pInf * (1.0 + 0.5 * (gam - 1.0) * m2) ** (gam / (gam - 1.0)) * (1.0 + 0.02 * rng.uniform(-1, 1, shp))
                f["ptInlet"] = np.asfortranarray(pt)
                f["ttInlet"] = np.asfortranarray(tt)
                f["htInlet"] = np.asfortranarray(gam / (gam - 1.0) * R * tt)
                d = -norm + 0.1 * rng.uniform(-1, 1, shp + (3,))          # roughly into the domain
                d /= np.sqrt((d ** 2).sum(axis=-1, keepdims=True))
                f["flowXdirInlet"] = np.asfortranarray(d[..., 0])
                f["flowYdirInlet"] = np.asfortranarray(d[..., 1])
                f["flowZdirInlet"] = np.asfortranarray(d[..., 2])
            else:
                f["subsonicInletTreatment"] = 2
                f["rho"] = np.asfortranarray(winf[0] * (1.0 + 0.02 * rng.uniform(-1, 1, shp)))
                vin = -0.5 * winf[1] * norm * (1.0 + 0.05 * rng.uniform(-1, 1, shp + (1,)))
                f["velx"] = np.asfortranarray(vin[..., 0])
                f["vely"] = np.asfortranarray(vin[..., 1])
                f["velz"] = np.asfortranarray(vin[..., 2])
        if typ in (-7, -8) and blk.nw > 5:
            f["turbInlet"] = np.asfortranarray(winf[5] * (1.0 + 0.3 * rng.uniform(0, 1, shp + (1,))))
        if typ == -7:
            f["rho"] = np.asfortranarray(winf[0] * (1.0 + 0.02 * rng.uniform(-1, 1, shp)))
            f["velx"] = np.asfortranarray(winf[1] * (1.0 + 0.02 * rng.uniform(-1, 1, shp)))
            f["vely"] = np.asfortranarray(winf[1] * 0.05 * rng.uniform(-1, 1, shp))
            f["velz"] = np.asfortranarray(winf[1] * 0.05 * rng.uniform(-1, 1, shp))
            f["ps"] = np.asfortranarray(prm.pInf * (1.0 + 0.02 * rng.uniform(-1, 1, shp)))
        faces.append(f)
    # cut the requested faces in two subfaces (ranges 1..h and h+1..end of the first index)
    for fid, typ2 in dict(split).items():
        m = next(ix for ix, f in enumerate(faces) if f["faceID"] == fid)
        f = faces[m]
        h = (f["icBeg"] + f["icEnd"]) // 2 if not split_at or fid not in split_at else int(split_at[fid])
        lo, hi = dict(f), dict(f)
        lo["icEnd"], hi["icBeg"], hi["bcType"] = h, h + 1, int(typ2)
        n0 = h - f["icBeg"] + 1
        for k in ("norm", "rface", "uSlip", "TNS_Wall", "rho", "velx", "vely", "velz", "ps", "ptInlet", "ttInlet", "htInlet",
                  "flowXdirInlet", "flowYdirInlet", "flowZdirInlet", "turbInlet"):
            if f.get(k) is not None:
                lo[k] = np.asfortranarray(f[k][:n0])
                hi[k] = np.asfortranarray(f[k][n0:])
        shp = hi["norm"].shape[:2]
        if typ2 in (-5, -6) and hi.get("rface") is None:
            hi["rface"] = np.asfortranarray(0.01 * rng.uniform(-1, 1, shp))
        if typ2 in (-3, -4) and hi.get("uSlip") is None:
            hi["uSlip"] = np.asfortranarray(0.02 * rng.uniform(-1, 1, shp + (3,)))
        if typ2 == -4 and hi.get("TNS_Wall") is None:
            tinf = prm.pInf / (prm.RGas * prm.rhoInf)
            hi["TNS_Wall"] = np.asfortranarray(tinf * (1.0 + 0.05 * rng.uniform(-1, 1, shp)))
        faces[m:m + 1] = [lo, hi]
    # viscous walls first (stable), as the reference orders the subfaces of a block
    faces.sort(key=lambda f: 0 if f["bcType"] in (-3, -4) else 1)
    nvisc = sum(1 for f in faces if f["bcType"] in (-3, -4))
    return faces, nvisc


def set_porosities(blk: Block, faces) -> None:
    """porI / porJ / porK of a block from its boundary subfaces, as setPorosities does (preprocessingAPI.F90:524-678): normalFlux
    everywhere, boundFlux on the faces of viscous / inviscid walls and extrapolation boundaries; every other boundary kind (farfield,
    symmetry, in- / outflow) and every 1-to-1 interface keeps normalFlux."""
    porI, porJ, porK = blk["porI"], blk["porJ"], blk["porK"]       # (1:il,2:jl,2:kl), (2:il,1:jl,2:kl), (2:il,2:jl,1:kl)
    porI[...] = normalFlux
    porJ[...] = normalFlux
    porK[...] = normalFlux
    for f in faces:
        if f["bcType"] not in (-3, -4, -5, -15):
            continue
        fid = f["faceID"]
        amax = blk.jl if fid <= 2 else blk.il
        bmax = blk.kl if fid <= 4 else blk.jl
        a0, a1 = max(f["icBeg"], 2) - 2, min(f["icEnd"], amax) - 2          # owned face cells, 0-based
        b0, b1 = max(f["jcBeg"], 2) - 2, min(f["jcEnd"], bmax) - 2
        if fid <= 2:
            porI[0 if fid == 1 else blk.nx, a0:a1 + 1, b0:b1 + 1] = boundFlux
        elif fid <= 4:
            porJ[a0:a1 + 1, 0 if fid == 3 else blk.ny, b0:b1 + 1] = boundFlux
        else:
            porK[a0:a1 + 1, b0:b1 + 1, 0 if fid == 5 else blk.nz] = boundFlux


# ----------------------------------------------------------------------------
# multigrid: coarsening maps of createCoarseBlocks (src/preprocessing/coarseUtils.F90:73-420)
# ----------------------------------------------------------------------------
def coarsen_1d(nf: int, state: str = "regular", keep=()):
    """One index direction of createCoarseBlocks for nf fine cells (nodes 1..nf+1).
    state: the fine level's iCoarsened ("regular" / "leftStarted" / "rightStarted"): a level that was itself coarsened
    irregularly from the left is swept from the right and vice versa (coarseUtils.F90:134-153).
    keep: fine NODES that must survive besides the block ends (the boundaries of the subfaces, :117-127).
    Every second node is dropped; where two kept nodes end up adjacent the coarse cell between them consists of ONE fine
    cell: its fine index is stored twice and its restriction weight is 1/2 (:281-295), and the fine cell interpolates from
    that coarse cell alone (:331-343).
    Returns (mgFine (1:ie_c,2), mgWeight (2:il_c), mgCoarse (2:il_f,2), nodeMap (coarse node 1..il_c -> fine node), state_c)."""
    assert nf >= 1
    iil = nf + 1
    co = np.zeros(iil + 2, bool)                # 1-based; co[0] and co[iil+1] unused
    co[1] = co[iil] = True
    for n in keep:
        co[int(n)] = True
    if state == "leftStarted":
        new = "rightStarted"
        for i in range(iil - 1, 1, -1):
            if not co[i + 1]:
                co[i] = True
    else:
        new = "leftStarted"
        for i in range(2, iil):
            if not co[i - 1]:
                co[i] = True
    il = int(co[1:iil + 1].sum())
    nc = il - 1
    if nf == 2 * nc:
        new = "regular"
    ie_c = nc + 2
    fine = np.zeros((ie_c, 2), np.int32, order="F")           # row m-1 = coarse cell m
    fine[0] = (0, 1)
    fine[ie_c - 1] = (nf + 2, nf + 3)
    weight = np.ones(nc, np.float64)
    coarse = np.zeros((nf, 2), np.int32, order="F")           # row i-2 = fine cell i
    ii = 2
    for i in range(2, iil + 1):
        if co[i]:
            if co[i - 1]:
                fine[ii - 1] = (i, i)
                weight[ii - 2] = 0.5
                coarse[i - 2] = (ii, ii)
            else:
                fine[ii - 1] = (i - 1, i)
                coarse[i - 2] = (ii, ii + 1)
            ii += 1
        else:
            coarse[i - 2] = (ii, ii - 1)
    node_map = np.array([i for i in range(1, iil + 1) if co[i]], np.int32)
    assert ii == il + 1 and node_map.size == il
    return fine, weight, coarse, node_map, new


def mg_maps_1d(nf: int):
    """Maps of one direction for nf fine cells coarsened from a regular level (even nf: 2:1, nc = nf/2).
    Returns (mgFine (1:ie_c,2), mgWeight (2:il_c), mgCoarse (2:il_f,2))."""
    return coarsen_1d(nf)[:3]


def make_coarse_block(fine: Block, prm: FlowParams, keep=((), (), ()), **mk) -> Block:
    """Level+1 block of `fine` with the transfer maps attached to both blocks.  Even cell counts: the same analytic map
    at half resolution; otherwise the coarse nodes ARE the surviving fine nodes (their parameters in the analytic map), halo
    nodes by extrapolation -- what the reference's coarse levels hold (coarseOwnedCoordinates, preprocessingAPI.F90:3945).
    keep: per direction the fine nodes where subfaces begin / end (they survive every coarsening, coarseUtils.F90:117-127).
    The coarse block records nodeMap = per direction the fine node of every coarse node 1..il."""
    maps = [coarsen_1d(nf, st, kp) for nf, st, kp in zip((fine.nx, fine.ny, fine.nz), fine.coarsened, keep)]
    ncs = [m[1].size for m in maps]
    regular = all((m[1] == 1.0).all() for m in maps)
    if fine.frame is not None:
        mk["frame"] = fine.frame
    if fine.frame is None and regular and (2 * ncs[0], 2 * ncs[1], 2 * ncs[2]) == (fine.nx, fine.ny, fine.nz):
        mk.pop("params", None)
        c = make_block(ncs[0], ncs[1], ncs[2], prm, **mk)
    else:
        fp = fine.nodeParams if fine.nodeParams is not None else node_params(fine.nx, fine.ny, fine.nz, mk.get("stretch_k", 1.0))
        par = []
        for t, m in zip(fp, maps):
            own = np.asarray(t)[m[3]]                       # parameters of the surviving nodes 1..il_c
            par.append(np.concatenate([[2.0 * own[0] - own[1]], own, [2.0 * own[-1] - own[-2]]]))
        mk["params"] = tuple(par)
        c = make_block(ncs[0], ncs[1], ncs[2], prm, **mk)
    c.coarsened = tuple(m[4] for m in maps)
    c.nodeMap = tuple(m[3] for m in maps)
    for d, m in zip("IJK", maps):
        c["mg%sFine" % d], c["mg%sWeight" % d] = m[0], m[1]
        fine["mg%sCoarse" % d] = m[2]
    ie, je, ke = c.ie, c.je, c.ke
    c["w1"] = F((ie, je, ke, 5))
    c["p1"] = F((ie, je, ke))
    c["wr"] = F((c.nx, c.ny, c.nz, 5))
    return c
