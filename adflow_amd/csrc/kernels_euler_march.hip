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

// MINW: minimum waves per SIMD requested from the register allocator (2: 256
// VGPRs, no spills; 3: 168 VGPRs; 4: 128 VGPRs) — selectable for A/B runs
// FW: persistent dissipation residual fw of the Runge-Kutta scheme is read/written
template <int MINW, bool FW>
__global__ __launch_bounds__(64 * EM_BY, MINW) void k_euler_march(const BlkView* __restrict__ tab, const int4* __restrict__ tiles,
                                                           KParams kp)
{
    const int4 t = tiles[blockIdx.x];           // x: block slot, y/z/w: tile coordinates
    if (t.x < 0) return;                        // padding entry of the XCD-ordered table
    const BlkView& b = tab[t.x];
    const int lane = threadIdx.x;
    const int i = t.y * EM_OUT + lane;          // columns i0-2 .. i0+61, i0 = 2 + 60*tx
    const int j = 2 + t.z * EM_BY + (int)threadIdx.y;
    const int k0 = 2 + t.w * EM_KCH;
    const int k1 = (k0 + EM_KCH - 1 < b.kl) ? k0 + EM_KCH - 1 : b.kl;
    const bool out = (lane >= 2 && lane <= 61 && i <= b.il && j <= b.jl);
    const int ic = (i < b.ib) ? i : b.ib, jc = (j < b.jl) ? j : b.jl;   // clamped: every lane takes part in the shuffles
    const long nb = b.nbox;
    const unsigned sj = 8u * (unsigned)b.ldi, sk = 8u * (unsigned)b.ldk;   // strides in bytes
    unsigned c = 8u * (unsigned)(ic + jc * b.ldi + k0 * b.ldk);           // byte offset of the column at plane k

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
    const unsigned fl_shift = 3;   // flags are bytes: element index = byte offset >> 3

    const double sslim = 0.001 * kp.pInfCorr;
    const double fis2 = kp.rFil * kp.vis2, fis4 = kp.rFil * kp.vis4;
    const bool doDiss = fabs(kp.rFil) >= 1.e-10;

    // 4-plane window k-2 .. k+1 of the own column
    Cell qm2 = ld_cell(m, c - 2 * sk), qm1 = ld_cell(m, c - sk), q0 = ld_cell(m, c);
    double radKm = ldg(radK, c - sk);
    int flagm = flags[(c - sk) >> fl_shift];                   // flags of cell k-1 (porK of the face below cell k)
    double dssKm = em_sensor(qm2.p, qm1.p, q0.p, sslim);   // k-sensor of cell k-1, carried along the march
    double accC[5] = {0, 0, 0, 0, 0}, accD[5] = {0, 0, 0, 0, 0};   // cell k-1: central / dissipative partial sums

    for (int k = k0; k <= k1 + 1; ++k) {
        const Cell qp1 = ld_cell(m, c + sk);
        const int flag0 = flags[c >> fl_shift];
        const double radK0 = ldg(radK, c);
        const double dssK0 = em_sensor(qm1.p, q0.p, qp1.p, sslim);
        // ---- k-face between cells k-1 and k: normal and porosity stored at cell k-1
        double fc[5], fd[5] = {0, 0, 0, 0, 0};
        {
            const double sx = ldg(sKx, c - sk), sy = ldg(sKy, c - sk), sz = ldg(sKz, c - sk);
            const int por = flg_porK((uint8_t)flagm);
            em_central(qm1, q0, sx, sy, sz, por, fc);
            if (doDiss) {
                const double rrad = (por == ADF_POR_NORMAL ? 0.5 : 0.0) * (radKm + radK0);
                em_jst(cons_of(qm2), cons_of(qm1), cons_of(q0), cons_of(qp1), rrad, dssKm, dssK0, fis2, fis4, fd);
            }
        }
        // ---- finish cell k-1 (left of the face) and write it
        if (k > k0 && out) {
            const unsigned cw = c - sk;
            const double blank = flg_blank((uint8_t)flagm);
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                double fwn = accD[l] - fd[l];
                if (FW) {
                    const double old = ldg(fw + l * nb, cw);
                    fwn = doDiss ? (kp.sfil * old + fwn) : old;
                    if (doDiss) stg(fw + l * nb, cw, fwn);
                }
                double d = accC[l] + fc[l];
                if (kp.coarseInit) d += ldg(wr + l * nb, cw);
                stg(dw + l * nb, cw, (d + fwn) * blank);
            }
        }
        if (k > k1) break;
        // ---- start cell k (right of the k-face)
#pragma unroll
        for (int l = 0; l < 5; ++l) { accC[l] = -fc[l]; accD[l] = fd[l]; }

        // ---- i-direction: this lane evaluates the face between i-1 and i, the face
        //      between i and i+1 comes from lane+1
        {
            const Cell qL = shfl_cell_up(q0, 1);
            const Cons W0 = cons_of(q0);
            const Cons WL = cons_of(qL);
            const Cons WLL = shfl_cons(W0, 2, true);
            const Cons WR = shfl_cons(W0, 1, false);
            const double sx = ldg(sIx, c - 8u), sy = ldg(sIy, c - 8u), sz = ldg(sIz, c - 8u);
            const int flagL = lane_up1(flag0);
            const int por = flg_porI((uint8_t)flagL);
            double gc[5], gd[5] = {0, 0, 0, 0, 0};
            em_central(qL, q0, sx, sy, sz, por, gc);
            if (doDiss) {
                const double pR = lane_dn1(q0.p);
                const double d0 = em_sensor(qL.p, q0.p, pR, sslim);
                const double dL = lane_up1(d0);
                const double rad0 = ldg(radI, c);
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
                    // (right cell += N, left cell -= N): one hand-over per component
                    const double n = gd[l] - gc[l];
                    accD[l] += n - lane_dn1(n);
                }
            }
        }
        // ---- j-direction: both faces of the cell, neighbours through L1/L2
        {
            const Cell qa = ld_cell(m, c - 2 * sj), qb = ld_cell(m, c - sj), qc = ld_cell(m, c + sj), qd = ld_cell(m, c + 2 * sj);
            const int porM = flg_porJ(flags[(c - sj) >> fl_shift]), porP = flg_porJ((uint8_t)flag0);
            double hc[5], hd[5];
            // minus face (j-1 | j): normal at cell j-1
            em_central(qb, q0, ldg(sJx, c - sj), ldg(sJy, c - sj), ldg(sJz, c - sj), porM, hc);
#pragma unroll
            for (int l = 0; l < 5; ++l) accC[l] -= hc[l];
            em_central(q0, qc, ldg(sJx, c), ldg(sJy, c), ldg(sJz, c), porP, hc);
#pragma unroll
            for (int l = 0; l < 5; ++l) accC[l] += hc[l];
            if (doDiss) {
                const double dm = em_sensor(qa.p, qb.p, q0.p, sslim), d0 = em_sensor(qb.p, q0.p, qc.p, sslim),
                             dp = em_sensor(q0.p, qc.p, qd.p, sslim);
                const double r0 = ldg(radJ, c);
                const double rrM = (porM == ADF_POR_NORMAL ? 0.5 : 0.0) * (ldg(radJ, c - sj) + r0);
                const double rrP = (porP == ADF_POR_NORMAL ? 0.5 : 0.0) * (r0 + ldg(radJ, c + sj));
                const Cons Wa = cons_of(qa), Wb = cons_of(qb), W0 = cons_of(q0), Wc = cons_of(qc), Wd = cons_of(qd);
                em_jst(Wa, Wb, W0, Wc, rrM, dm, d0, fis2, fis4, hd);
#pragma unroll
                for (int l = 0; l < 5; ++l) accD[l] += hd[l];
                em_jst(Wb, W0, Wc, Wd, rrP, d0, dp, fis2, fis4, hd);
#pragma unroll
                for (int l = 0; l < 5; ++l) accD[l] -= hd[l];
            }
        }
        // ---- advance the window
        qm2 = qm1; qm1 = q0; q0 = qp1;
        radKm = radK0;
        dssKm = dssK0;
        flagm = flag0;
        c += sk;
    }
}

int g_march_minw = 2;

void launch_euler_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s)
{
    if (ntiles <= 0) return;
    const dim3 blk(64, EM_BY, 1);
    if (kp.fwMode) {
        hipLaunchKernelGGL((k_euler_march<2, true>), dim3(ntiles), blk, 0, s, tab, tiles, kp);
    } else if (g_march_minw >= 3) {
        hipLaunchKernelGGL((k_euler_march<3, false>), dim3(ntiles), blk, 0, s, tab, tiles, kp);
    } else {
        hipLaunchKernelGGL((k_euler_march<2, false>), dim3(ntiles), blk, 0, s, tab, tiles, kp);
    }
}

// tile decomposition of one block for the table built by the host
void euler_march_tiles(const BlkView& b, int* ntx, int* nty, int* ntz)
{
    *ntx = (b.nx + EM_OUT - 1) / EM_OUT;
    *nty = (b.ny + EM_BY - 1) / EM_BY;
    *ntz = (b.nz + EM_KCH - 1) / EM_KCH;
}
