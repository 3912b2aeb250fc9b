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
