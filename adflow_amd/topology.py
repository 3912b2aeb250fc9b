"""Block topologies and 1-to-1 halo communication patterns.

Builds, for synthetic multi-block cases, exactly the data the reference's
preprocessing produces for the hot path: `internalCell_{1st,2nd}(level)` and
`commPatternCell_{1st,2nd}(level)` (src/modules/communication.F90, built in
src/preprocessing/pointMatchedCommPattern.F90 — out of scope).
Cell indices are 0..ib as the reference stores them; block ids are local and
1-based; ranks are 0-based.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Tuple

import numpy as np


@dataclass
class CommPattern:
    """Flattened pattern of ONE rank, one level, one halo depth."""
    donorBlock: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    donorIndices: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.int32, order="F"))
    haloBlock: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    haloIndices: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.int32, order="F"))
    sendProc: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    nsendCum: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))
    sendBlock: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    sendIndices: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.int32, order="F"))
    recvProc: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    nrecvCum: np.ndarray = field(default_factory=lambda: np.zeros(1, np.int32))
    recvBlock: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    recvIndices: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.int32, order="F"))
    # per same-process copy: -1 / +1 when the halo wraps around the brick in i (a periodic interface), else 0
    wrapI: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))

    @property
    def ncopy(self):
        return int(self.donorBlock.size)


@dataclass
class BrickTopology:
    """Bi x Bj x Bk equal blocks of nx x ny x nz cells, periodic in the index
    directions flagged in `periodic` (default: all three; the halos beyond a
    non-periodic end of the brick belong to boundary subfaces and appear in no
    pattern); block g (0-based, i fastest) lives on rank owner(g)."""
    Bi: int
    Bj: int
    Bk: int
    nx: int
    ny: int
    nz: int
    owner: Callable[[int], int] = lambda g: 0
    periodic: Tuple[bool, bool, bool] = (True, True, True)

    @property
    def nblocks(self):
        return self.Bi * self.Bj * self.Bk

    def gid(self, bi, bj, bk):
        return bi + self.Bi * (bj + self.Bj * bk)

    def coords(self, g):
        return g % self.Bi, (g // self.Bi) % self.Bj, g // (self.Bi * self.Bj)

    def local_ids(self) -> Dict[int, int]:
        """global block id -> local 1-based block number nn on its owner."""
        cnt: Dict[int, int] = {}
        out = {}
        for g in range(self.nblocks):
            r = self.owner(g)
            cnt[r] = cnt.get(r, 0) + 1
            out[g] = cnt[r]
        return out

    def blocks_of(self, rank) -> List[int]:
        return [g for g in range(self.nblocks) if self.owner(g) == rank]

    def boundary_spec(self, g, brick_spec: Dict[int, int]) -> Dict[int, int]:
        """{faceID: BCType} of block g: the faces of the block that lie on a NON-periodic end of the brick take the kind
        `brick_spec` gives for that end of the brick (faceID 1..6 = iMin, iMax, jMin, jMax, kMin, kMax); every other face of
        the block is a 1-to-1 interface."""
        c, B = self.coords(g), (self.Bi, self.Bj, self.Bk)
        out = {}
        for d in range(3):
            if self.periodic[d]:
                continue
            if c[d] == 0 and (2 * d + 1) in brick_spec:
                out[2 * d + 1] = brick_spec[2 * d + 1]
            if c[d] == B[d] - 1 and (2 * d + 2) in brick_spec:
                out[2 * d + 2] = brick_spec[2 * d + 2]
        return out

    def patterns(self, nLayers: int, only_rank=None) -> Dict[int, CommPattern]:
        """CommPattern per rank for halo depth nLayers (1: cells 1..ie, 2: 0..ib),
        faces, edges and corners included.  only_rank: build just that rank's
        pattern (skips block pairs that do not involve it).
        nLayers = 0: the NODE pattern (commPatternNode_1st / internalNode_1st): halo nodes 0 and ie/je/ke take the
        coordinates of the neighbour's interior nodes nx / 2 (exchangeCoor, haloExchange.F90:2456)."""
        nx, ny, nz = self.nx, self.ny, self.nz
        nodes = nLayers == 0
        if nodes:
            ii, jj, kk = np.arange(0, nx + 3), np.arange(0, ny + 3), np.arange(0, nz + 3)
            I, J, K = np.meshgrid(ii, jj, kk, indexing="ij")
            halo = (I == 0) | (I == nx + 2) | (J == 0) | (J == ny + 2) | (K == 0) | (K == nz + 2)
        else:
            lo = 2 - nLayers
            ii = np.arange(lo, nx + 2 + nLayers)
            jj = np.arange(lo, ny + 2 + nLayers)
            kk = np.arange(lo, nz + 2 + nLayers)
            I, J, K = np.meshgrid(ii, jj, kk, indexing="ij")
            halo = ~((I >= 2) & (I <= nx + 1) & (J >= 2) & (J <= ny + 1) & (K >= 2) & (K <= nz + 1))
        hi, hj, hk = I[halo], J[halo], K[halo]        # canonical order: k slowest? (C order of the mask)
        lid = self.local_ids()
        ranks = sorted({self.owner(g) for g in range(self.nblocks)})
        loc = {r: [[], [], [], [], []] for r in ranks}        # donorBlk, donorIdx, haloBlk, haloIdx, wrapI
        msg: Dict[Tuple[int, int], List] = {}                  # (src rank, dst rank) -> [sendBlk, sendIdx, recvBlk, recvIdx]

        def node_map(b, idx, n, B):
            # node idx of block b in one direction: halo nodes 0 / n+2 are nodes n / 2 of the previous / next block
            g_ = (b * n + idx - 1) % (B * n)
            keep = (idx >= 1) & (idx <= n + 1)
            return np.where(keep, b, g_ // n), np.where(keep, idx, g_ % n + 1)

        for g in range(self.nblocks):
            bi, bj, bk = self.coords(g)
            if nodes:
                gw = bi * nx + hi - 1
                wrap = np.where((hi >= 1) & (hi <= nx + 1), 0, np.where(gw < 0, -1, np.where(gw >= self.Bi * nx, 1, 0)))
                dbi, di = node_map(bi, hi, nx, self.Bi)
                dbj, dj = node_map(bj, hj, ny, self.Bj)
                dbk, dk = node_map(bk, hk, nz, self.Bk)
                dg = dbi + self.Bi * (dbj + self.Bj * dbk)
            else:
                gw = bi * nx + hi - 2
                wrap = np.where(gw < 0, -1, np.where(gw >= self.Bi * nx, 1, 0))
                gi = (bi * nx + hi - 2) % (self.Bi * nx)
                gj = (bj * ny + hj - 2) % (self.Bj * ny)
                gk = (bk * nz + hk - 2) % (self.Bk * nz)
                dg = (gi // nx) + self.Bi * ((gj // ny) + self.Bj * (gk // nz))
                di, dj, dk = gi % nx + 2, gj % ny + 2, gk % nz + 2
            # halos beyond a non-periodic end of the brick have no donor
            if nodes:
                gwj, gwk = bj * ny + hj - 1, bk * nz + hk - 1
                inside = [(gw >= 0) & (gw <= self.Bi * nx), (gwj >= 0) & (gwj <= self.Bj * ny), (gwk >= 0) & (gwk <= self.Bk * nz)]
            else:
                gwj, gwk = bj * ny + hj - 2, bk * nz + hk - 2
                inside = [(gw >= 0) & (gw < self.Bi * nx), (gwj >= 0) & (gwj < self.Bj * ny), (gwk >= 0) & (gwk < self.Bk * nz)]
            keepm = np.ones(hi.shape, bool)
            for d in range(3):
                if not self.periodic[d]:
                    keepm &= inside[d]
            rh = self.owner(g)
            udg = np.unique(dg[keepm])
            downers = np.array([self.owner(int(x)) for x in udg])
            for dgu, rd in zip(udg, downers):
                if only_rank is not None and rh != only_rank and rd != only_rank:
                    continue
                m = (dg == dgu) & keepm
                didx = np.stack([di[m], dj[m], dk[m]], axis=1)
                hidx = np.stack([hi[m], hj[m], hk[m]], axis=1)
                n = int(m.sum())
                if rd == rh:
                    L = loc[rh]
                    L[0].append(np.full(n, lid[int(dgu)])); L[1].append(didx)
                    L[2].append(np.full(n, lid[g])); L[3].append(hidx); L[4].append(wrap[m])
                else:
                    M = msg.setdefault((int(rd), int(rh)), [[], [], [], []])
                    M[0].append(np.full(n, lid[int(dgu)])); M[1].append(didx)
                    M[2].append(np.full(n, lid[g])); M[3].append(hidx)
        out = {}
        for r in ranks:
            if only_rank is not None and r != only_rank:
                continue
            cp = CommPattern()
            L = loc[r]
            if L[0]:
                cp.donorBlock = np.concatenate(L[0]).astype(np.int32)
                cp.donorIndices = np.asfortranarray(np.concatenate(L[1]).astype(np.int32))
                cp.haloBlock = np.concatenate(L[2]).astype(np.int32)
                cp.haloIndices = np.asfortranarray(np.concatenate(L[3]).astype(np.int32))
                cp.wrapI = np.concatenate(L[4]).astype(np.int32)
            sp, sc, sb, si_ = [], [0], [], []
            rp, rc, rb, ri_ = [], [0], [], []
            for (src, dst), M in sorted(msg.items()):
                if src == r:
                    sp.append(dst); sb.append(np.concatenate(M[0])); si_.append(np.concatenate(M[1]))
                    sc.append(sc[-1] + sb[-1].size)
                if dst == r:
                    rp.append(src); rb.append(np.concatenate(M[2])); ri_.append(np.concatenate(M[3]))
                    rc.append(rc[-1] + rb[-1].size)
            if sp:
                cp.sendProc = np.array(sp, np.int32); cp.nsendCum = np.array(sc, np.int32)
                cp.sendBlock = np.concatenate(sb).astype(np.int32)
                cp.sendIndices = np.asfortranarray(np.concatenate(si_).astype(np.int32))
            if rp:
                cp.recvProc = np.array(rp, np.int32); cp.nrecvCum = np.array(rc, np.int32)
                cp.recvBlock = np.concatenate(rb).astype(np.int32)
                cp.recvIndices = np.asfortranarray(np.concatenate(ri_).astype(np.int32))
            out[r] = cp
        return out


def apply_local_copies(blocks: Dict[int, object], cp: CommPattern, names=("w", "p", "rlv", "rev")):
    """numpy statement of the same-process copies (haloExchange.F90:657-678) used
    to give synthetic multi-block states consistent halos before the first step."""
    for t in range(cp.ncopy):
        db, hb = blocks[int(cp.donorBlock[t])], blocks[int(cp.haloBlock[t])]
        di, dj, dk = cp.donorIndices[t]
        hi, hj, hk = cp.haloIndices[t]
        for n in names:
            if n in db.a and n in hb.a:
                hb.a[n][hi, hj, hk] = db.a[n][di, dj, dk]


def apply_local_copies_fast(blocks: Dict[int, object], cp: CommPattern, names=("w", "p", "rlv", "rev")):
    """Vectorised version of apply_local_copies (grouped by block pair)."""
    if cp.ncopy == 0:
        return
    key = cp.donorBlock.astype(np.int64) * 100000 + cp.haloBlock
    for kv in np.unique(key):
        m = key == kv
        db, hb = blocks[int(kv // 100000)], blocks[int(kv % 100000)]
        d, h = cp.donorIndices[m], cp.haloIndices[m]
        for n in names:
            if n in db.a and n in hb.a:
                hb.a[n][h[:, 0], h[:, 1], h[:, 2]] = db.a[n][d[:, 0], d[:, 1], d[:, 2]]


# ----------------------------------------------------------------------------
# general 1-to-1 topologies: blocks of different sizes joined with any orientation
# ----------------------------------------------------------------------------
@dataclass
class LatticeBlock:
    """One block of a LatticeTopology: `dims` = (nx, ny, nz) cells; `T` = the 3x3 signed permutation that turns a step in the
    block's (i, j, k) into a step in the global lattice (column d = the lattice direction of the block's index direction d: the
    transformation the reference stores per subface as l1, l2, l3, modules/block.F90:271-309, here per block);
    `origin` = the lattice cell of the block's first owned cell (2, 2, 2)."""
    dims: Tuple[int, int, int]
    T: np.ndarray
    origin: Tuple[int, int, int]


class LatticeTopology:
    """Blocks embedded in ONE global lattice of unit cells, each with its own orientation and size: every face that two blocks
    share is a 1-to-1 interface with the transformation matrix T_a^-1 T_b between their index systems, faces on the outside of
    the union are physical boundaries.  Produces what the reference's preprocessing (pointMatchedCommPattern.F90, indirectHalos.F90:
    out of scope) hands to the hot path: internalCell / commPatternCell lists of the 1st and 2nd halo layer incl. the halos
    reached across an edge or corner, and the node pattern of exchangeCoor.  The interface of BrickTopology where the checks need it."""

    def __init__(self, blocks: List[LatticeBlock], owner: Callable[[int], int] = lambda g: 0, stretch_z: float = 1.0, amp: float = 0.02):
        self.blocks = blocks
        self.owner = owner
        self.stretch_z = stretch_z
        self.amp = amp
        lo = np.full(3, 10 ** 9)
        hi = np.full(3, -10 ** 9)
        for b in blocks:
            T = np.asarray(b.T, int)
            assert abs(round(np.linalg.det(T))) == 1 and (np.abs(T).sum(axis=0) == 1).all() and (np.abs(T).sum(axis=1) == 1).all()
            c0 = np.asarray(b.origin)
            c1 = c0 + T @ (np.asarray(b.dims) - 1)
            lo = np.minimum(lo, np.minimum(c0, c1))
            hi = np.maximum(hi, np.maximum(c0, c1))
        self.lo, self.hi = lo, hi
        self.size = hi - lo + 1
        self.grid = np.full(tuple(self.size), -1, np.int32)         # owner block of every lattice cell
        for g, b in enumerate(blocks):
            c = self._cells(g, *np.meshgrid(*[np.arange(2, n + 2) for n in b.dims], indexing="ij"))
            assert (self.grid[c[0] - lo[0], c[1] - lo[1], c[2] - lo[2]] == -1).all(), "blocks overlap"
            self.grid[c[0] - lo[0], c[1] - lo[1], c[2] - lo[2]] = g

    # ---- BrickTopology's interface ----
    @property
    def nblocks(self):
        return len(self.blocks)

    def dims(self, g):
        return tuple(self.blocks[g].dims)

    def local_ids(self) -> Dict[int, int]:
        cnt: Dict[int, int] = {}
        out = {}
        for g in range(self.nblocks):
            r = self.owner(g)
            cnt[r] = cnt.get(r, 0) + 1
            out[g] = cnt[r]
        return out

    def blocks_of(self, rank) -> List[int]:
        return [g for g in range(self.nblocks) if self.owner(g) == rank]

    def with_owner(self, owner):
        return LatticeTopology(self.blocks, owner, self.stretch_z, self.amp)

    # ---- index maps ----
    def _cells(self, g, i, j, k):
        """lattice cells of the cells (i, j, k) of block g"""
        b = self.blocks[g]
        T = np.asarray(b.T, int)
        d = np.stack([np.asarray(i) - 2, np.asarray(j) - 2, np.asarray(k) - 2])
        return [b.origin[a] + T[a, 0] * d[0] + T[a, 1] * d[1] + T[a, 2] * d[2] for a in range(3)]

    def _node_origin(self, g):
        """lattice POSITION of node (1, 1, 1) of block g (the lower corner of cell (2, 2, 2) in the block's own orientation):
        the cell occupies [c, c + 1] of the lattice; along a reversed direction its lower corner is c + 1"""
        b = self.blocks[g]
        T = np.asarray(b.T, int)
        return np.asarray(b.origin) + (T.sum(axis=1) < 0).astype(int)

    def _owner_of(self, c):
        inside = np.ones(np.asarray(c[0]).shape, bool)
        idx = []
        for a in range(3):
            ca = np.asarray(c[a]) - self.lo[a]
            inside &= (ca >= 0) & (ca < self.size[a])
            idx.append(np.clip(ca, 0, self.size[a] - 1))
        return np.where(inside, self.grid[idx[0], idx[1], idx[2]], -1)

    def _to_block(self, g, c):
        """cell indices (i, j, k) in block g of the lattice cells c"""
        b = self.blocks[g]
        T = np.asarray(b.T, int)
        d = [np.asarray(c[a]) - b.origin[a] for a in range(3)]
        return [T[0, m] * d[0] + T[1, m] * d[1] + T[2, m] * d[2] + 2 for m in range(3)]       # T^-1 = T^T

    def frame(self, g, level_scale=1):
        """what make_nodes needs to evaluate the ONE analytic map of the whole mesh on block g's nodes (adflow_amd/synth.py):
        lattice position = o + T (local offsets from node 1), global parameter = position / extent of the lattice"""
        return dict(T=np.asarray(self.blocks[g].T, float), o=self._node_origin(g).astype(float) - self.lo,
                    scale=1.0 / self.size.astype(float), stretch_z=self.stretch_z, amp=self.amp)

    def make_block(self, g, prm, seed=1, **mk):
        from .synth import make_block
        nx, ny, nz = self.blocks[g].dims
        mk.pop("stretch_k", None)
        mk.setdefault("amp", self.amp)
        params = tuple(np.arange(n + 3) - 1.0 for n in (nx, ny, nz))        # lattice offsets of the nodes 0..ie from node 1
        return make_block(nx, ny, nz, prm, seed=seed, params=params, frame=self.frame(g), **mk)

    def coarse(self, dims_of=None):
        """the topology of the next multigrid level (2:1 in every direction: all dims and origins even)"""
        cb = []
        for b in self.blocks:
            assert all(n % 2 == 0 for n in b.dims)
            T = np.asarray(b.T, int)
            # the first owned coarse cell holds the fine cells (2, 3): its lattice cell is floor(min(c(2), c(3)) / 2)
            c2 = np.asarray(b.origin) - self.lo
            c3 = c2 + T.sum(axis=1)
            cb.append(LatticeBlock(tuple(n // 2 for n in b.dims), b.T, tuple(int(v) for v in np.minimum(c2, c3) // 2)))
        assert all(v % 2 == 0 for v in self.size)
        return LatticeTopology(cb, self.owner, self.stretch_z, self.amp)

    def boundary_spec(self, g, spec: Dict[int, int]) -> Dict[int, int]:
        """{faceID of block g: BCType} for the faces of g on the OUTSIDE of the mesh; `spec` is keyed by the outward LATTICE direction
        (1..6 = -x, +x, -y, +y, -z, +z).  A face is either an interface or a boundary as a whole here."""
        b = self.blocks[g]
        T = np.asarray(b.T, int)
        out = {}
        for d in range(3):
            for side in (0, 1):
                rng = [np.arange(2, n + 2) for n in b.dims]
                rng[d] = np.array([1 if side == 0 else b.dims[d] + 2])
                own = self._owner_of(self._cells(g, *np.meshgrid(*rng, indexing="ij")))
                if (own >= 0).all():
                    continue
                assert (own < 0).all(), "a block face must be an interface or a boundary as a whole"
                v = T[:, d] * (-1 if side == 0 else 1)                 # outward lattice direction
                a = int(np.nonzero(v)[0][0])
                key = 2 * a + (2 if v[a] > 0 else 1)
                if key in spec:
                    out[2 * d + 1 + side] = spec[key]
        return out

    def patterns(self, nLayers: int, only_rank=None) -> Dict[int, CommPattern]:
        """CommPattern per rank (see BrickTopology.patterns); nLayers = 0: the node pattern."""
        nodes = nLayers == 0
        lid = self.local_ids()
        ranks = sorted({self.owner(g) for g in range(self.nblocks)})
        loc = {r: [[], [], [], []] for r in ranks}
        msg: Dict[Tuple[int, int], List] = {}
        for g, b in enumerate(self.blocks):
            nx, ny, nz = b.dims
            if nodes:
                I, J, K = np.meshgrid(np.arange(0, nx + 3), np.arange(0, ny + 3), np.arange(0, nz + 3), indexing="ij")
                halo = (I == 0) | (I == nx + 2) | (J == 0) | (J == ny + 2) | (K == 0) | (K == nz + 2)
            else:
                lo = 2 - nLayers
                I, J, K = np.meshgrid(np.arange(lo, nx + 2 + nLayers), np.arange(lo, ny + 2 + nLayers), np.arange(lo, nz + 2 + nLayers),
                                      indexing="ij")
                halo = ~((I >= 2) & (I <= nx + 1) & (J >= 2) & (J <= ny + 1) & (K >= 2) & (K <= nz + 1))
            h = [I[halo], J[halo], K[halo]]
            if nodes:
                # the cell the halo node is a corner of: the halo cell in the directions where the node lies outside 1..il, an owned
                # cell elsewhere; its owner is the donor block, the donor node the same lattice point in the donor's numbering
                cell = [np.where(h[d] == 0, 1, np.where(h[d] == b.dims[d] + 2, b.dims[d] + 2, np.clip(h[d], 2, b.dims[d] + 1)))
                        for d in range(3)]
                dg = self._owner_of(self._cells(g, *cell))
                T = np.asarray(b.T, int)
                pos = [self._node_origin(g)[a] + T[a, 0] * (h[0] - 1) + T[a, 1] * (h[1] - 1) + T[a, 2] * (h[2] - 1) for a in range(3)]
            else:
                c = self._cells(g, *h)
                dg = self._owner_of(c)
            rh = self.owner(g)
            for dgu in np.unique(dg[dg >= 0]):
                dgu = int(dgu)
                assert dgu != g
                rd = self.owner(dgu)
                if only_rank is not None and rh != only_rank and rd != only_rank:
                    continue
                m = dg == dgu
                if nodes:
                    db = self.blocks[dgu]
                    Td = np.asarray(db.T, int)
                    dpos = [pos[a][m] - self._node_origin(dgu)[a] for a in range(3)]
                    didx = np.stack([Td[0, q] * dpos[0] + Td[1, q] * dpos[1] + Td[2, q] * dpos[2] + 1 for q in range(3)], axis=1)
                    assert all(((didx[:, q] >= 1) & (didx[:, q] <= db.dims[q] + 1)).all() for q in range(3))
                else:
                    didx = np.stack(self._to_block(dgu, [c[a][m] for a in range(3)]), axis=1)
                hidx = np.stack([h[0][m], h[1][m], h[2][m]], axis=1)
                n = int(m.sum())
                if rd == rh:
                    L = loc[rh]
                    L[0].append(np.full(n, lid[dgu])); L[1].append(didx)
                    L[2].append(np.full(n, lid[g])); L[3].append(hidx)
                else:
                    M = msg.setdefault((int(rd), int(rh)), [[], [], [], []])
                    M[0].append(np.full(n, lid[dgu])); M[1].append(didx)
                    M[2].append(np.full(n, lid[g])); M[3].append(hidx)
        out = {}
        for r in ranks:
            if only_rank is not None and r != only_rank:
                continue
            cp = CommPattern()
            L = loc[r]
            if L[0]:
                cp.donorBlock = np.concatenate(L[0]).astype(np.int32)
                cp.donorIndices = np.asfortranarray(np.concatenate(L[1]).astype(np.int32))
                cp.haloBlock = np.concatenate(L[2]).astype(np.int32)
                cp.haloIndices = np.asfortranarray(np.concatenate(L[3]).astype(np.int32))
                cp.wrapI = np.zeros(cp.donorBlock.size, np.int32)
            sp, sc, sb, si_ = [], [0], [], []
            rp, rc, rb, ri_ = [], [0], [], []
            for (src, dst), M in sorted(msg.items()):
                if src == r:
                    sp.append(dst); sb.append(np.concatenate(M[0])); si_.append(np.concatenate(M[1]))
                    sc.append(sc[-1] + sb[-1].size)
                if dst == r:
                    rp.append(src); rb.append(np.concatenate(M[2])); ri_.append(np.concatenate(M[3]))
                    rc.append(rc[-1] + rb[-1].size)
            if sp:
                cp.sendProc = np.array(sp, np.int32); cp.nsendCum = np.array(sc, np.int32)
                cp.sendBlock = np.concatenate(sb).astype(np.int32)
                cp.sendIndices = np.asfortranarray(np.concatenate(si_).astype(np.int32))
            if rp:
                cp.recvProc = np.array(rp, np.int32); cp.nrecvCum = np.array(rc, np.int32)
                cp.recvBlock = np.concatenate(rb).astype(np.int32)
                cp.recvIndices = np.asfortranarray(np.concatenate(ri_).astype(np.int32))
            out[r] = cp
        return out


def ell_topology(scale: int = 1, owner: Callable[[int], int] = lambda g: 0, stretch_z: float = 1.0, fill: bool = True) -> LatticeTopology:
    """Four blocks of different sizes and orientations that fill a 36 x 16 x 14 box (scale 1: 24x16x8, 16x12x8, 16x6x24, 6x16x12):
    A: (i, j, k) = (+x, +y, +z) at the lattice origin;
    B: beyond A's iMax face, joined through ITS jMin face: (i, j, k) = (-y, +x, +z) -- A's j runs against B's i;
    C: on top of A's kMax face, joined through its jMin face: (i, j, k) = (+y, +z, +x) -- a cyclic permutation;
    D: above B and beside C: (i, j, k) = (+z, +y, +x) -- a LEFT-handed block (blockType%rightHanded = F).
    B and C (and A and D) meet along an edge only: each has edge halos whose donor is the other one (indirect halos).
    fill = False leaves D out: an L-shaped mesh whose re-entrant edge has halos without donor or boundary condition (only
    entry points that never read edge halos are meaningful there)."""
    s = scale
    A = LatticeBlock((24 * s, 16 * s, 8 * s), np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1]]), (0, 0, 0))
    B = LatticeBlock((16 * s, 12 * s, 8 * s), np.array([[0, 1, 0], [-1, 0, 0], [0, 0, 1]]), (24 * s, 16 * s - 1, 0))
    C = LatticeBlock((16 * s, 6 * s, 24 * s), np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]]), (0, 0, 8 * s))
    D = LatticeBlock((6 * s, 16 * s, 12 * s), np.array([[0, 0, 1], [0, 1, 0], [1, 0, 0]]), (24 * s, 0, 8 * s))
    return LatticeTopology([A, B, C] + ([D] if fill else []), owner, stretch_z)
