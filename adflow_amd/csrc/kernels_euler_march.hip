// Euler residual, central flux + scalar JST dissipation — k-marching kernel.
//
// Same arithmetic as the generic gather kernel (kernels_inviscid.hip; reference
// fluxes.F90:4-401 central, :1049-1436 scalar JST, residuals.F90:334-344 sum)
// reorganised for CDNA4:
//   * one launch covers EVERY block of a multigrid level (tile table), so small
//     blocks still fill 256 CUs and the launch count does not grow with nDom;
//   * a workgroup is 64 lanes (i) x EM_BY rows (j); every thread marches along k
//     through a chunk of EM_KCH cells holding a 4-plane window of the state in
//     registers, so each k-face flux is evaluated ONCE and the state of a column
//     is fetched once per chunk;
//   * a wavefront covers 64 consecutive i-columns of which the inner 60 produce
//     output: every i-face flux is evaluated ONCE by the lane right of it and
//     handed to the left neighbour with a wave shuffle (no LDS, no atomics);
//   * the j-direction is a gather through L1/L2 (both j-faces per cell).
// Per cell: 4 face evaluations instead of 6 and ~48 loads instead of ~100.
// Roofline: HBM, 175 B/cell algorithmic.
#include "internal.h"

#define EM_OUT 60
#define EM_BY 4
#define EM_KCH 32

struct Cell { double rho, u, v, w, e, p; };
struct Cons { double r, ru, rv, rw, ep; };   // rho, rho*u, rho*v, rho*w, rhoE + p

// uniform (SGPR-resident) base pointers of one block with 32-bit element indices:
// lets the compiler emit global_load with scalar base + 32-bit vector offset
struct EmPtrs {
    GPTR(const double) w0; GPTR(const double) w1; GPTR(const double) w2; GPTR(const double) w3; GPTR(const double) w4;
    GPTR(const double) p;
};

__device__ __forceinline__ Cell ld_cell(const EmPtrs& m, unsigned o)   // o: byte offset
{
    Cell q;
    q.rho = ldg(m.w0, o); q.u = ldg(m.w1, o); q.v = ldg(m.w2, o); q.w = ldg(m.w3, o); q.e = ldg(m.w4, o);
    q.p = ldg(m.p, o);
    return q;
}

__device__ __forceinline__ Cons cons_of(const Cell& q)
{
    Cons c;
    c.r = q.rho; c.ru = q.u * q.rho; c.rv = q.v * q.rho; c.rw = q.w * q.rho; c.ep = q.e + q.p;
    return c;
}

__device__ __forceinline__ Cell shfl_cell_up(const Cell& q, int d)
{
    Cell r;
    (void)d;   // d == 1
    r.rho = lane_up1(q.rho); r.u = lane_up1(q.u); r.v = lane_up1(q.v);
    r.w = lane_up1(q.w); r.e = lane_up1(q.e); r.p = lane_up1(q.p);
    return r;
}

__device__ __forceinline__ Cons shfl_cons(const Cons& q, int d, bool up)
{
    Cons r;
    if (up) {   // d == 2: two single-lane shifts
        (void)d;
        r.r = lane_up1(lane_up1(q.r)); r.ru = lane_up1(lane_up1(q.ru)); r.rv = lane_up1(lane_up1(q.rv));
        r.rw = lane_up1(lane_up1(q.rw)); r.ep = lane_up1(lane_up1(q.ep));
    } else {    // d == 1
        r.r = lane_dn1(q.r); r.ru = lane_dn1(q.ru); r.rv = lane_dn1(q.rv);
        r.rw = lane_dn1(q.rw); r.ep = lane_dn1(q.ep);
    }
    return r;
}

__device__ __forceinline__ Cons shfl_cons_up1(const Cons& q)
{
    Cons r;
    r.r = lane_up1(q.r); r.ru = lane_up1(q.ru); r.rv = lane_up1(q.rv); r.rw = lane_up1(q.rw); r.ep = lane_up1(q.ep);
    return r;
}

// central flux through the face between L and R (fluxes.F90:52-129);  dw(L) += f, dw(R) -= f
__device__ __forceinline__ void em_central(const Cell& L, const Cell& R, double sx, double sy, double sz, int por, double f[5])
{
    double vnp = R.u * sx + R.v * sy + R.w * sz;
    double vnm = L.u * sx + L.v * sy + L.w * sz;
    double porVel = 1.0, porFlux = 0.5;
    if (por == ADF_POR_NOFLUX) porFlux = 0.0;
    if (por == ADF_POR_BOUND) { porVel = 0.0; vnp = 0.0; vnm = 0.0; }
    porVel *= porFlux;
    const double qsp = vnp * porVel, qsm = vnm * porVel;
    const double rqsp = qsp * R.rho, rqsm = qsm * L.rho;
    const double pa = porFlux * (R.p + L.p);
    f[0] = rqsp + rqsm;
    f[1] = rqsp * R.u + rqsm * L.u + pa * sx;
    f[2] = rqsp * R.v + rqsm * L.v + pa * sy;
    f[3] = rqsp * R.w + rqsm * L.w + pa * sz;
    f[4] = qsp * R.e + qsm * L.e + porFlux * (vnp * R.p + vnm * L.p);
}

// scalar JST flux through the face between L and R (fluxes.F90:1204-1272);  fw(R) += f, fw(L) -= f
__device__ __forceinline__ void em_jst(const Cons& LL, const Cons& L, const Cons& R, const Cons& RR, double rrad, double dssL,
                                       double dssR, double fis2, double fis4, double f[5])
{
    const double dis2 = fis2 * rrad * fmin(0.25, fmax(dssL, dssR));
    const double dis4 = fmax(fis4 * rrad - dis2, 0.0);
    double ddw;
    ddw = R.r - L.r;   f[0] = dis2 * ddw - dis4 * (RR.r - LL.r - 3.0 * ddw);
    ddw = R.ru - L.ru; f[1] = dis2 * ddw - dis4 * (RR.ru - LL.ru - 3.0 * ddw);
    ddw = R.rv - L.rv; f[2] = dis2 * ddw - dis4 * (RR.rv - LL.rv - 3.0 * ddw);
    ddw = R.rw - L.rw; f[3] = dis2 * ddw - dis4 * (RR.rw - LL.rw - 3.0 * ddw);
    ddw = R.ep - L.ep; f[4] = dis2 * ddw - dis4 * (RR.ep - LL.ep - 3.0 * ddw);
}

__device__ __forceinline__ double em_sensor(double sm, double s0, double sp, double sslim)
{
    return fabs((sp - 2.0 * s0 + sm) / (sp + 2.0 * s0 + sm + sslim));
}

// ---------------------------------------------------------------------------
// Software-pipelined form of the same march.  One plane of the march is split
// in two phases, each with its own set of global loads:
//   A(k): k-face (k-1|k), finish + store cell k-1, i-face   (16 loads)
//   B(k): both j-faces of cell k                               (32 loads)
// and the loads of a phase are issued BEFORE the arithmetic of the other
// phase, so every load has ~half an iteration of FP64 work (plus the other
// waves of the SIMD) to hide its latency instead of being waited for
// immediately.  The register allocator gets the full 256-VGPR budget (2 waves
// per SIMD); latency hiding comes from the pipeline, not from occupancy.
// ---------------------------------------------------------------------------
// spectral radii of one cell from its state and the SUMS of its two face normals per direction: the arithmetic of k_time_step
// (solverUtils.F90:131-199, blocks at rest), evaluated inside the march by the RADII form of the kernel below
struct Rad3 { double rI, rJ, rK; };

__device__ __forceinline__ Rad3 em_radii(const KParams& kp, const Cell& q, const double sIs[3], const double sJs[3], const double sKs[3])
{
    const double clim2 = 0.000001 * kp.gammaInf * kp.pInfCorr / kp.rhoInf;
    const double cc2 = fmax(kp.gammaConstant * q.p * rcp_nr(q.rho), clim2);
    const double si2 = sIs[0] * sIs[0] + sIs[1] * sIs[1] + sIs[2] * sIs[2];
    const double sj2 = sJs[0] * sJs[0] + sJs[1] * sJs[1] + sJs[2] * sJs[2];
    const double sk2 = sKs[0] * sKs[0] + sKs[1] * sKs[1] + sKs[2] * sKs[2];
    double ri = 0.5 * (fabs(q.u * sIs[0] + q.v * sIs[1] + q.w * sIs[2]) + kp.acousticScaleFactor * fastsqrt(cc2 * si2));
    double rj = 0.5 * (fabs(q.u * sJs[0] + q.v * sJs[1] + q.w * sJs[2]) + kp.acousticScaleFactor * fastsqrt(cc2 * sj2));
    double rk = 0.5 * (fabs(q.u * sKs[0] + q.v * sKs[1] + q.w * sKs[2]) + kp.acousticScaleFactor * fastsqrt(cc2 * sk2));
    Rad3 r;
    if (kp.doScaling) {
        // radI = ri (1 + (rj/ri)^adis + (rk/ri)^adis) etc. (solverUtils.F90:187-199): three powers of the radii themselves
        // (fast_powa, internal.h) and three reciprocals instead of the three log + three exp of k_time_step
        const double epsr = 1.e-25;
        ri = fmax(ri, epsr); rj = fmax(rj, epsr); rk = fmax(rk, epsr);
        const double x = fast_powa(ri, kp.adis), y = fast_powa(rj, kp.adis), z = fast_powa(rk, kp.adis);
        r.rI = ri * (1.0 + (y + z) * rcp_nr(x));
        r.rJ = rj * (1.0 + (z + x) * rcp_nr(y));
        r.rK = rk * (1.0 + (x + y) * rcp_nr(z));
    } else {
        r.rI = ri; r.rJ = rj; r.rK = rk;
    }
    return r;
}

struct InA {      // loads of phase A of plane k (the state of cell k+1 goes straight into the window)
    double radK0;             // radK(k)
    double sKx, sKy, sKz;     // normal of the face k-1|k (stored at k-1)
    double sIx, sIy, sIz;     // normal of the face i-1|i (stored at i-1)
    double radI0;             // radI(i)
    double fwOld[5];          // FW only: fw of cell k-1
    int flag0;                // flags of cell k
};
struct InB {      // loads of phase B of plane k
    Cell qa, qb, qc, qd;      // j-2, j-1, j+1, j+2
    double sMx, sMy, sMz;     // normal of the face j-1|j
    double sPx, sPy, sPz;     // normal of the face j|j+1
    double rJm, rJ0, rJp;     // radJ at j-1, j, j+1
    int flagJm;               // flags of cell j-1
};

// LDSJ: the state rows j0-2 .. j0+EM_BY+1 of a plane are fetched ONCE per workgroup
// (16-byte loads, two rows per wave instruction) into a 3-slot LDS ring and shared by
// the EM_BY waves: 6 wide loads per wave and plane replace 30 narrow ones, which takes
// the texture-address unit (the measured bottleneck of the register-only form:
// TA busy 83-95 % of the kernel) out of the critical path.  One barrier per plane.
typedef double d2_t __attribute__((vector_size(16)));
__device__ __forceinline__ d2_t ldg2(GPTR(const double) base, unsigned byteoff)
{
    return *(GPTR(const d2_t))((GPTR(const char))base + byteoff);
}

// RADII (with LDSJ): the spectral radii are formed inside the march instead of being read from radI/J/K:
// the radii of cell plane k+1 are computed at the end of step k -- own cell from the normals in flight plus nine loads the next
// steps repeat anyway (L1 / L2 hits), the two waves at the tile edges also the cell of the row outside the tile -- and radJ goes
// to the neighbouring rows through a double-buffered LDS slot behind the barrier the state ring has anyway.  Removes the
// k_time_step pass in front of an evaluation that does not need dtl (blocketteRes without updateIntermed).
template <bool FW, bool LDSJ, int BY, bool RADII = false>
__global__ __launch_bounds__(64 * BY, 2) void k_euler_march_p(const BlkView* __restrict__ tab, const int4* __restrict__ tiles,
                                                                 KParams kp, int kch)
{
    // BY rows of cells per workgroup (4: two workgroups per CU; 8: one, with 12 instead of 2 x 8 staged rows per 8 produced)
    constexpr int EL_ROWS = BY + 4, EL_PANEL = EL_ROWS * 64, EL_SLOT = 6 * EL_PANEL;
    __shared__ __attribute__((aligned(16))) double lds[LDSJ ? 3 * EL_SLOT : 2];
    __shared__ double rjx[RADII ? 2 * (BY + 2) * 64 : 1];       // radJ of the rows j0-1 .. j0+BY: [plane parity][row][lane]
    const int4 t = tiles[blockIdx.x];
    if (t.x < 0) return;
    const BlkView& b = tab[t.x];
    const int lane = threadIdx.x;
    const int i = t.y * EM_OUT + lane;
    // RADII: the waves 0 and BY-1 of a tile also form the radii of the rows outside it; every second workgroup shifts its
    // wave -> row assignment by BY / 2 so that those waves of two co-resident workgroups sit on different SIMDs
    const int ty = RADII ? wave_uniform((int)((threadIdx.y + (BY / 2) * (blockIdx.x & 1u)) % BY)) : (int)threadIdx.y;
    const int j = 2 + t.z * BY + ty;
    const int k0 = 2 + t.w * kch;
    const int k1 = (k0 + kch - 1 < b.kl) ? k0 + kch - 1 : b.kl;
    const bool out = (lane >= 2 && lane <= 61 && i <= b.il && j <= b.jl);
    const int ic = (i < b.ib) ? i : b.ib, jc = (j < b.jl) ? j : b.jl;
    const long nb = b.nbox;
    const unsigned sj = 8u * (unsigned)b.ldi, sk = 8u * (unsigned)b.ldk;
    unsigned c = 8u * (unsigned)(ic + jc * b.ldi + k0 * b.ldk);

    EmPtrs m;
    m.w0 = (GPTR(const double))b.w; m.w1 = m.w0 + nb; m.w2 = m.w1 + nb; m.w3 = m.w2 + nb; m.w4 = m.w3 + nb;
    m.p = (GPTR(const double))b.p;
    GPTR(const double) radI = (GPTR(const double))b.radI;
    GPTR(const double) radJ = (GPTR(const double))b.radJ;
    GPTR(const double) radK = (GPTR(const double))b.radK;
    GPTR(const double) sIx = (GPTR(const double))b.sI; GPTR(const double) sIy = sIx + nb; GPTR(const double) sIz = sIy + nb;
    GPTR(const double) sJx = (GPTR(const double))b.sJ; GPTR(const double) sJy = sJx + nb; GPTR(const double) sJz = sJy + nb;
    GPTR(const double) sKx = (GPTR(const double))b.sK; GPTR(const double) sKy = sKx + nb; GPTR(const double) sKz = sKy + nb;
    GPTR(const uint8_t) flags = (GPTR(const uint8_t))b.flags;
    GPTR(double) dw = (GPTR(double))b.dw;
    GPTR(double) fw = (GPTR(double))b.fw;
    GPTR(const double) wr = (GPTR(const double))b.wr;

    const double sslim = 0.001 * kp.pInfCorr;
    const double fis2 = kp.rFil * kp.vis2, fis4 = kp.rFil * kp.vis4;
    const bool doDiss = fabs(kp.rFil) >= 1.e-10;

    struct RawN { double sI[3], sJm[3], sJ[3], sKm[3], sK[3]; };     // RADII: normals requested for the radii of the next plane
    RawN rawOwn, rawEdge;
    InA a;
    // phase-A loads of the plane at byte offset cc; the state of the cell above it lands in `up`
    auto loadA = [&](unsigned cc, Cell& up) {
        if (!LDSJ) up = ld_cell(m, cc + sk);
        a.flag0 = flags[cc >> 3];
        if (!RADII) a.radK0 = ldg(radK, cc);
        a.sKx = ldg(sKx, cc - sk); a.sKy = ldg(sKy, cc - sk); a.sKz = ldg(sKz, cc - sk);
        a.sIx = ldg(sIx, cc - 8u); a.sIy = ldg(sIy, cc - 8u); a.sIz = ldg(sIz, cc - 8u);
        if (!RADII) a.radI0 = ldg(radI, cc);
        if (FW) {
#pragma unroll
            for (int l = 0; l < 5; ++l) a.fwOld[l] = ldg(fw + l * nb, cc - sk);
        }
    };
    auto loadB = [&](unsigned cc, InB& q) {
        if (!LDSJ) {
            q.qb = ld_cell(m, cc - sj); q.qc = ld_cell(m, cc + sj);
            q.qa = ld_cell(m, cc - 2 * sj); q.qd = ld_cell(m, cc + 2 * sj);
        }
        q.flagJm = flags[(cc - sj) >> 3];
        q.sMx = ldg(sJx, cc - sj); q.sMy = ldg(sJy, cc - sj); q.sMz = ldg(sJz, cc - sj);
        q.sPx = ldg(sJx, cc); q.sPy = ldg(sJy, cc); q.sPz = ldg(sJz, cc);
        if (!RADII) { q.rJm = ldg(radJ, cc - sj); q.rJ0 = ldg(radJ, cc); q.rJp = ldg(radJ, cc + sj); }
    };

    // ---- LDS ring (LDSJ): slot of plane k0+n is n % 3.  This wave stages rows 2*ty and
    //      2*ty+1 of the 8-row panel: lanes 0-31 the first, lanes 32-63 the second row,
    //      two adjacent cells per lane.
    const int hrow = 2 * ty + (lane >> 5), xp = 2 * (lane & 31);
    unsigned cs = 0;            // byte offset of this lane's pair in the plane being staged
    int splane = k0;            // plane index of the next staging load
    d2_t st[6];
    if (LDSJ) {
        int jr = t.z * BY + hrow;                  // j0 - 2 + hrow
        if (jr > b.jb) jr = b.jb;
        int ip = t.y * EM_OUT + xp;                   // even; the last pair may run one cell past ib (row padding)
        if (ip > (b.ib & ~1)) ip = b.ib & ~1;         // stays 16-byte aligned
        cs = 8u * (unsigned)(ip + jr * b.ldi);
    }
    const int ldsW = hrow * 64 + xp;                  // position inside a panel
    const bool stager = hrow < EL_ROWS;               // BY = 8: the waves 6 and 7 have no rows to stage
    auto stage_load = [&]() {
        if (!stager) return;
        const int pl = (splane < b.kb) ? splane : b.kb;
        const unsigned o = cs + (unsigned)pl * sk;
        st[0] = ldg2(m.w0, o); st[1] = ldg2(m.w1, o); st[2] = ldg2(m.w2, o); st[3] = ldg2(m.w3, o); st[4] = ldg2(m.w4, o);
        st[5] = ldg2(m.p, o);
        ++splane;
    };
    auto stage_store = [&](int slot) {
        if (!stager) return;
#pragma unroll
        for (int q = 0; q < 6; ++q) *(d2_t*)&lds[slot * EL_SLOT + q * EL_PANEL + ldsW] = st[q];
    };
    auto lds_cell = [&](int slot, int row) {
        const double* __restrict__ r = &lds[slot * EL_SLOT + row * 64 + lane];
        Cell q;
        q.rho = r[0]; q.u = r[EL_PANEL]; q.v = r[2 * EL_PANEL]; q.w = r[3 * EL_PANEL]; q.e = r[4 * EL_PANEL];
        q.p = r[5 * EL_PANEL];
        return q;
    };
    int sl0 = 0, sl1 = 1, sl2 = 2;   // slots of the planes k, k+1, k+2

    // prologue: window k-2 .. k of the own column, phase-A loads of the first plane
    Cell S0 = ld_cell(m, c - 2 * sk), S1 = ld_cell(m, c - sk), S2 = ld_cell(m, c), S3;
    // RADII: the face normals of a cell (sI of the face i-1|i -- the face i|i+1 comes from lane+1 --, both sJ, both sK) are requested
    // one phase before the radii are formed from them, so that their latency hides behind the barrier and the next phase
    auto raw_issue = [&](unsigned cc, RawN& n) {
        n.sI[0] = ldg(sIx, cc - 8u); n.sI[1] = ldg(sIy, cc - 8u); n.sI[2] = ldg(sIz, cc - 8u);
        n.sJm[0] = ldg(sJx, cc - sj); n.sJm[1] = ldg(sJy, cc - sj); n.sJm[2] = ldg(sJz, cc - sj);
        n.sJ[0] = ldg(sJx, cc); n.sJ[1] = ldg(sJy, cc); n.sJ[2] = ldg(sJz, cc);
        n.sKm[0] = ldg(sKx, cc - sk); n.sKm[1] = ldg(sKy, cc - sk); n.sKm[2] = ldg(sKz, cc - sk);
        n.sK[0] = ldg(sKx, cc); n.sK[1] = ldg(sKy, cc); n.sK[2] = ldg(sKz, cc);
    };
    auto raw_radii = [&](const RawN& n, const Cell& q) {
        double sIs[3], sJs[3], sKs[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) { sIs[d] = n.sI[d] + lane_dn1(n.sI[d]); sJs[d] = n.sJm[d] + n.sJ[d]; sKs[d] = n.sKm[d] + n.sK[d]; }
        return em_radii(kp, q, sIs, sJs, sKs);
    };
    // offsets of the cells of the rows outside the tile (clamped like the staged rows), same column and plane as c
    const int jlo = 1 + t.z * BY, jhi_ = 2 + t.z * BY + BY;                    // j0 - 1, j0 + BY
    const unsigned cLo = 8u * (unsigned)(ic + ((jlo < b.jb) ? jlo : b.jb) * b.ldi + k0 * b.ldk);
    const unsigned cHi = 8u * (unsigned)(ic + ((jhi_ < b.jb) ? jhi_ : b.jb) * b.ldi + k0 * b.ldk);
    // the own cell at its TRUE row (clamped at the end of the box only: a row beyond jl still publishes the radJ of the real cell
    // je, which the row below needs for its upper face), state from the ring
    const unsigned cOwn = 8u * (unsigned)(ic + ((j < b.jb) ? j : b.jb) * b.ldi + k0 * b.ldk);
    const unsigned cEdge = (ty == 0) ? cLo : cHi;          // the row outside the tile this wave looks after (waves 0 and BY-1)
    const bool edgeWave = (ty == 0 || ty == BY - 1);
    const int edgeRowSlot = (ty == 0) ? 1 : BY + 2, edgeOut = (ty == 0) ? 0 : BY + 1;
    int pl = 0;                      // current plane relative to k0
    auto radii_issue = [&](int q) {
        const unsigned koff = (unsigned)q * sk;
        raw_issue(cOwn + koff, rawOwn);
        if (edgeWave) raw_issue(cEdge + koff, rawEdge);
    };
    // radii of plane k0 + q from the requested normals and the state ring slot of that plane: own row -> `own` and rjx slot ty + 1;
    // waves 0 / BY-1 also the row outside the tile.  (BY = 1 is not instantiated: one wave would own both outside rows.)
    auto radii_finish = [&](int q, int slotPlane, Rad3& own) {
        const int par = q & 1;
        own = raw_radii(rawOwn, lds_cell(slotPlane, ty + 2));
        rjx[(par * (BY + 2) + ty + 1) * 64 + lane] = own.rJ;
        if (edgeWave) rjx[(par * (BY + 2) + edgeOut) * 64 + lane] = raw_radii(rawEdge, lds_cell(slotPlane, edgeRowSlot)).rJ;
    };
    Rad3 Rcur;                       // RADII: radii of the own cell at the current plane
    Rcur.rI = Rcur.rJ = Rcur.rK = 0.0;
    double radKm = 0.0;
    if (RADII) {
        RawN n;
        raw_issue(c - sk, n);
        radKm = raw_radii(n, S1).rK;
    } else
        radKm = ldg(radK, c - sk);
    int flagm = flags[(c - sk) >> 3];
    if (LDSJ) {
        stage_load(); stage_store(0);
        stage_load(); stage_store(1);
        stage_load();                 // plane k0+2: stays in flight until the first step stores it
    }
    if (RADII) radii_issue(0);
    loadA(c, S3);
    Rad3 Rnext = Rcur;
    if (RADII) {
        __syncthreads();              // plane k0 of the state ring is complete
        radii_finish(0, 0, Rcur);
        radii_issue(1);               // plane k0+1: formed during the first step
    }
    if (LDSJ) {
        __syncthreads();
        S3 = lds_cell(1, ty + 2);
    }
    double dssKm = em_sensor(S0.p, S1.p, S2.p, sslim);
    double accC[5] = {0, 0, 0, 0, 0}, accD[5] = {0, 0, 0, 0, 0};
    double dssK0;

    // k-face (k-1|k) from the phase-A loads; finishes and stores cell k-1, starts cell k
    auto kface = [&](bool store, const Cell& qm2, const Cell& qm1, const Cell& q0, const Cell& qp1) {
        dssK0 = em_sensor(qm1.p, q0.p, qp1.p, sslim);
        double fc[5], fd[5] = {0, 0, 0, 0, 0};
        const int por = flg_porK((uint8_t)flagm);
        em_central(qm1, q0, a.sKx, a.sKy, a.sKz, por, fc);
        if (doDiss) {
            const double rrad = (por == ADF_POR_NORMAL ? 0.5 : 0.0) * (radKm + (RADII ? Rcur.rK : a.radK0));
            em_jst(cons_of(qm2), cons_of(qm1), cons_of(q0), cons_of(qp1), rrad, dssKm, dssK0, fis2, fis4, fd);
        }
        if (store && out) {
            const unsigned cw = c - sk;
            const double blank = flg_blank((uint8_t)flagm);
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                double fwn = accD[l] - fd[l];
                if (FW) {
                    const double old = a.fwOld[l];
                    fwn = doDiss ? (kp.sfil * old + fwn) : old;
                    if (doDiss) stg(fw + l * nb, cw, fwn);
                }
                double d = accC[l] + fc[l];
                if (kp.coarseInit) d += ldg(wr + l * nb, cw);
                stg(dw + l * nb, cw, (d + fwn) * blank);
            }
        }
#pragma unroll
        for (int l = 0; l < 5; ++l) { accC[l] = -fc[l]; accD[l] = fd[l]; }
    };

    // one plane of the march; qm2's slot receives the state of cell k+2 (phase-A loads of plane k+1)
    auto plane = [&](bool first, Cell& qm2, const Cell& qm1, const Cell& q0, const Cell& qp1) {
        // ---- issue the loads of phase B(k); they complete under the arithmetic of A(k)
        InB q;
        if (LDSJ) {
            stage_store(sl2);         // plane k+2 (loaded during the previous step)
            stage_load();             // plane k+3
        }
        loadB(c, q);
        // RADII: radii of plane k+1 from the normals requested at the end of the previous step (state: ring slot of plane k+1);
        // radJ goes to the other parity of rjx, read by the next step behind the barrier that closes this one
        if (RADII) radii_finish(pl + 1, sl1, Rnext);
        __builtin_amdgcn_sched_barrier(0);

        // ================= phase A(k) =================
        kface(!first, qm2, qm1, q0, qp1);
        const int flag0 = a.flag0;
        const double radK0 = RADII ? Rcur.rK : a.radK0;
        {
            const Cell qL = shfl_cell_up(q0, 1);
            const Cons W0 = cons_of(q0);
            const Cons WL = cons_of(qL);
            const Cons WLL = shfl_cons_up1(WL);
            const Cons WR = shfl_cons(W0, 1, false);
            const int flagL = lane_up1(flag0);
            const int por = flg_porI((uint8_t)flagL);
            double gc[5], gd[5] = {0, 0, 0, 0, 0};
            em_central(qL, q0, a.sIx, a.sIy, a.sIz, por, gc);
            if (doDiss) {
                const double pR = lane_dn1(q0.p);
                const double d0 = em_sensor(qL.p, q0.p, pR, sslim);
                const double dL = lane_up1(d0);
                const double rad0 = RADII ? Rcur.rI : a.radI0;
                const double radL = lane_up1(rad0);
                const double rrad = (por == ADF_POR_NORMAL ? 0.5 : 0.0) * (radL + rad0);
                em_jst(WLL, WL, W0, WR, rrad, dL, d0, fis2, fis4, gd);
            }
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                if (FW) {
                    const double gcP = lane_dn1(gc[l]);
                    const double gdP = lane_dn1(gd[l]);
                    accC[l] += gcP - gc[l];      // + plus face, - minus face
                    accD[l] += gd[l] - gdP;      // fw(R) += f : minus face adds, plus face subtracts
                } else {
                    // no persistent fw: only the net contribution N = D - F matters
                    const double n = gd[l] - gc[l];
                    accD[l] += n - lane_dn1(n);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- issue the loads of phase A(k+1); they complete under the arithmetic of B(k)
        loadA(c + sk, qm2);
        __builtin_amdgcn_sched_barrier(0);

        // ================= phase B(k) =================
        if (LDSJ) {
            q.qa = lds_cell(sl0, ty); q.qb = lds_cell(sl0, ty + 1); q.qc = lds_cell(sl0, ty + 3); q.qd = lds_cell(sl0, ty + 4);
        }
        {
            const int porM = flg_porJ((uint8_t)q.flagJm), porP = flg_porJ((uint8_t)flag0);
            double hc[5], hd[5];
            em_central(q.qb, q0, q.sMx, q.sMy, q.sMz, porM, hc);
#pragma unroll
            for (int l = 0; l < 5; ++l) accC[l] -= hc[l];
            em_central(q0, q.qc, q.sPx, q.sPy, q.sPz, porP, hc);
#pragma unroll
            for (int l = 0; l < 5; ++l) accC[l] += hc[l];
            if (doDiss) {
                const double dm = em_sensor(q.qa.p, q.qb.p, q0.p, sslim), d0 = em_sensor(q.qb.p, q0.p, q.qc.p, sslim),
                             dp = em_sensor(q0.p, q.qc.p, q.qd.p, sslim);
                if (RADII) {
                    const int par = pl & 1;
                    q.rJm = rjx[(par * (BY + 2) + ty) * 64 + lane];
                    q.rJ0 = Rcur.rJ;
                    q.rJp = rjx[(par * (BY + 2) + ty + 2) * 64 + lane];
                }
                const double rrM = (porM == ADF_POR_NORMAL ? 0.5 : 0.0) * (q.rJm + q.rJ0);
                const double rrP = (porP == ADF_POR_NORMAL ? 0.5 : 0.0) * (q.rJ0 + q.rJp);
                const Cons Wa = cons_of(q.qa), Wb = cons_of(q.qb), W0 = cons_of(q0), Wc = cons_of(q.qc), Wd = cons_of(q.qd);
                em_jst(Wa, Wb, W0, Wc, rrM, dm, d0, fis2, fis4, hd);
#pragma unroll
                for (int l = 0; l < 5; ++l) accD[l] += hd[l];
                em_jst(Wb, W0, Wc, Wd, rrP, d0, dp, fis2, fis4, hd);
#pragma unroll
                for (int l = 0; l < 5; ++l) accD[l] -= hd[l];
            }
        }
        radKm = radK0;
        dssKm = dssK0;
        flagm = flag0;
        c += sk;
        if (RADII) {
            ++pl;
            Rcur = Rnext;
            radii_issue(pl + 1);      // normals of plane k+2: in flight across the barrier
        }
        if (LDSJ) {
            // plane k+2 is complete in LDS, plane k is no longer read: its slot takes plane k+3
            __syncthreads();
            qm2 = lds_cell(sl2, ty + 2);      // own column at k+2: the next step's qp1
            const int s = sl0; sl0 = sl1; sl1 = sl2; sl2 = s;
        }
    };

    for (int k = k0; k <= k1; ++k) {
        plane(k == k0, S0, S1, S2, S3);
        const Cell nxt = S0;
        S0 = S1; S1 = S2; S2 = S3; S3 = nxt;
    }
    // epilogue: face k1|k1+1 closes the last cell of the chunk
    kface(true, S0, S1, S2, S3);
}

int g_march_kch = EM_KCH;      // k-chunk length of a tile (tuning "march_kch")
static int march_rows() { return EM_BY; }

void launch_euler_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s)
{
    if (ntiles <= 0) return;
    const dim3 blk(64, EM_BY, 1);
    if (kp.fwMode)
        hipLaunchKernelGGL((k_euler_march_p<true, true, EM_BY>), dim3(ntiles), blk, 0, s, tab, tiles, kp, g_march_kch);
    else if (kp.radiiInMarch)
        hipLaunchKernelGGL((k_euler_march_p<false, true, EM_BY, true>), dim3(ntiles), blk, 0, s, tab, tiles, kp, g_march_kch);
    else
        hipLaunchKernelGGL((k_euler_march_p<false, true, EM_BY>), dim3(ntiles), blk, 0, s, tab, tiles, kp, g_march_kch);
}

// true when launch_euler_march would run the form of the kernel that can compute the radii itself
bool euler_march_radii_capable(const KParams& kp)
{
    // fast_powa is built for moderate exponents (2^(adis log2 r) must stay far from the ends of the exponent range)
    return !kp.fwMode && (!kp.doScaling || (kp.adis > 0.0 && kp.adis <= 2.0));
}

// tile decomposition of one block for the table built by the host
void euler_march_tiles(const BlkView& b, int* ntx, int* nty, int* ntz)
{
    *ntx = (b.nx + EM_OUT - 1) / EM_OUT;
    *nty = (b.ny + march_rows() - 1) / march_rows();
    *ntz = (b.nz + g_march_kch - 1) / g_march_kch;
}
