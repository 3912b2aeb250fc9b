// Spectral radii + local time step  (reference: solverUtils::timeStep_block,
// src/solver/solverUtils.F90:43-356) and the JST entropy sensor variable
// (fluxes.F90:1113-1138) as ONE cell-parallel pass.
//
// Roofline: HBM.  Algorithmic bytes/cell: read w(4 of 5) p gamma 9 face-normal
// components [rlv rev vol], write radI radJ radK dtl [ss]  (DESIGN.md §4).
#include "internal.h"

#define TS_BX 64
#define TS_BY 4

template <bool VISC, bool SCALING>
__global__ __launch_bounds__(TS_BX* TS_BY) void k_time_step(const BlkView* __restrict__ tab, int nzb, KParams kp)
{
    // level-batched: blockIdx.z = (block slot, plane); slots the level does not use hold an all-zero view
    const BlkView& b = tab[blockIdx.z / nzb + 1];
    if (b.nx == 0) return;
    // lane 0 of the first block sits at i = 2-16 so that every wavefront load
    // starts on a 128-byte line (see internal.h)
    const int i = blockIdx.x * TS_BX + threadIdx.x + (2 - 16);
    const int j = blockIdx.y * TS_BY + threadIdx.y;
    const int k = blockIdx.z % nzb;
    if (k > b.kb) return;
    if (i < 0 || i > b.ib || j > b.jb) return;
    const long c = b.idx(i, j, k);
    const long nb = b.nbox;

    const double rho = b.w[c];
    const double pp = b.p[c];
    const double gam = b.gamma[c];

    if (VISC && !kp.dissApprox) {
        // entropy-like sensor variable, all cells 0..ib (fluxes.F90:1126-1136); left alone while the sensor is frozen
        b.ss[c] = pp / pow(rho, gam);
    }
    if (i < 1 || i > b.ie || j < 1 || j > b.je || k < 1 || k > b.ke) return;

    const double clim2 = 0.000001 * kp.gammaInf * kp.pInfCorr / kp.rhoInf;
    const double uux = b.w[c + nb], uuy = b.w[c + 2 * nb], uuz = b.w[c + 3 * nb];
    double cc2 = gam * pp / rho;
    cc2 = fmax(cc2, clim2);

    // sums of the two face normals of the cell in every direction: from the stored normals, or (tuning metric_from_x & 4) re-formed
    // from the eight corner nodes with the formulas of metric_block (adjointExtra.F90:176-268) -- 24 B of coordinates per cell from
    // HBM instead of 72 B of normals, the corner nodes are shared by eight cells
    double sIs[3], sJs[3], sKs[3];
    if (kp.metricFromX & 4) {
        const long sj = b.ldi, sk = b.ldk;
        double n[2][2][2][3];                       // [di][dj][dk]: x(i-1+di, j-1+dj, k-1+dk)
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
            for (int dj = 0; dj < 2; ++dj)
#pragma unroll
                for (int dk = 0; dk < 2; ++dk) {
                    const long q = c - (1 - di) - (1 - dj) * sj - (1 - dk) * sk;
#pragma unroll
                    for (int d = 0; d < 3; ++d) n[di][dj][dk][d] = b.x[q + d * nb];
                }
        auto cross_add = [&](const double* p1, const double* p2, const double* q1, const double* q2, double* sN) {
            const double v1x = p1[0] - p2[0], v1y = p1[1] - p2[1], v1z = p1[2] - p2[2];
            const double v2x = q1[0] - q2[0], v2y = q1[1] - q2[1], v2z = q1[2] - q2[2];
            sN[0] += b.mfact * (v1y * v2z - v1z * v2y);
            sN[1] += b.mfact * (v1z * v2x - v1x * v2z);
            sN[2] += b.mfact * (v1x * v2y - v1y * v2x);
        };
#pragma unroll
        for (int d = 0; d < 3; ++d) sIs[d] = sJs[d] = sKs[d] = 0.0;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            // sI at node plane i-1+f: v1 = x(i,j,n) - x(i,m,k) ; v2 = x(i,j,k) - x(i,m,n)
            cross_add(n[f][1][0], n[f][0][1], n[f][1][1], n[f][0][0], sIs);
            // sJ at node row j-1+f: v1 = x(i,j,n) - x(l,j,k) ; v2 = x(l,j,n) - x(i,j,k)
            cross_add(n[1][f][0], n[0][f][1], n[0][f][0], n[1][f][1], sJs);
            // sK at node plane k-1+f: v1 = x(i,j,k) - x(l,m,k) ; v2 = x(l,j,k) - x(i,m,k)
            cross_add(n[1][1][f], n[0][0][f], n[0][1][f], n[1][0][f], sKs);
        }
    } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            sIs[d] = b.sI[c - 1 + d * nb] + b.sI[c + d * nb];
            sJs[d] = b.sJ[c - b.ldi + d * nb] + b.sJ[c + d * nb];
            sKs[d] = b.sK[c - b.ldk + d * nb] + b.sK[c + d * nb];
        }
    }
    // i-direction: faces i-1 and i
    double sx = sIs[0], sy = sIs[1], sz = sIs[2];
    const double si2 = sx * sx + sy * sy + sz * sz;
    // grid velocity of a moving block: sum over the two faces (solverUtils.F90:147-181)
    double sFace = 0.0, sFaceJ = 0.0, sFaceK = 0.0;
    if (b.sFace) {      // uniform branch: no loads at all for blocks at rest
        const adf_real8* sF = b.sFace;
        sFace = sF[c - 1] + sF[c];
        sFaceJ = sF[c - b.ldi + nb] + sF[c + nb];
        sFaceK = sF[c - b.ldk + 2 * nb] + sF[c + 2 * nb];
    }
    double ri = 0.5 * (fabs(uux * sx + uuy * sy + uuz * sz - sFace) + kp.acousticScaleFactor * sqrt(cc2 * si2));

    sx = sJs[0]; sy = sJs[1]; sz = sJs[2];
    const double sj2 = sx * sx + sy * sy + sz * sz;
    sFace = sFaceJ;
    double rj = 0.5 * (fabs(uux * sx + uuy * sy + uuz * sz - sFace) + kp.acousticScaleFactor * sqrt(cc2 * sj2));

    sx = sKs[0]; sy = sKs[1]; sz = sKs[2];
    const double sk2 = sx * sx + sy * sy + sz * sz;
    sFace = sFaceK;
    double rk = 0.5 * (fabs(uux * sx + uuy * sy + uuz * sz - sFace) + kp.acousticScaleFactor * sqrt(cc2 * sk2));

    const double rsum = ri + rj + rk;   // inviscid part of 1/dt (before scaling)

    if (SCALING) {
        const double epsr = 1.e-25;
        ri = fmax(ri, epsr);
        rj = fmax(rj, epsr);
        rk = fmax(rk, epsr);
        // (ri/rj)**adis etc. (solverUtils.F90:187-199) through one logarithm per
        // radius: x**a = exp(a (log xi - log xj)); the three FP64 pow calls of the
        // straightforward form made this kernel VALU-bound (profiles/r01_b_*).
        const double li = log(ri), lj = log(rj), lk = log(rk);
        const double rij = exp(kp.adis * (li - lj));
        const double rjk = exp(kp.adis * (lj - lk));
        const double rki = exp(kp.adis * (lk - li));
        b.radI[c] = ri * (1.0 + 1.0 / rij + rki);
        b.radJ[c] = rj * (1.0 + 1.0 / rjk + rij);
        b.radK[c] = rk * (1.0 + 1.0 / rki + rjk);
    } else {
        b.radI[c] = ri;
        b.radJ[c] = rj;
        b.radK[c] = rk;
    }

    if (kp.onlyRadii) return;
    // halo cells keep the inviscid sum (reference stores it there too)
    double dt = rsum;
    const bool owned = (i >= 2 && i <= b.il && j >= 2 && j <= b.jl && k >= 2 && k <= b.kl);
    if (owned) {
        if (VISC) {
            double rmu = b.rlv[c];
            if (kp.eddyModel) rmu += b.rev[c];
            rmu = 0.5 * rmu / (rho * b.vol[c]);
            dt += rmu * si2;
            dt += rmu * sj2;
            dt += rmu * sk2;
        }
        const double plim = 0.001 * kp.pInfCorr;
        const double p0 = pp;
        double pa = b.p[c + 1], pb = b.p[c - 1];
        const double dpi = fabs(pa - 2.0 * p0 + pb) / (pa + 2.0 * p0 + pb + plim);
        pa = b.p[c + b.ldi];
        pb = b.p[c - b.ldi];
        const double dpj = fabs(pa - 2.0 * p0 + pb) / (pa + 2.0 * p0 + pb + plim);
        pa = b.p[c + b.ldk];
        pb = b.p[c - b.ldk];
        const double dpk = fabs(pa - 2.0 * p0 + pb) / (pa + 2.0 * p0 + pb + plim);
        const double rfl = 1.0 / (1.0 + 2.0 * (dpi + dpj + dpk));
        dt = rfl / dt;
    }
    b.dtl[c] = dt;
}

// JST entropy sensor variable alone, all cells 0..ib (fluxes.F90:1126-1136):
// used when the state changed since the last time-step pass.
__global__ __launch_bounds__(TS_BX* TS_BY) void k_entropy(BlkView b)
{
    const int i = blockIdx.x * TS_BX + threadIdx.x + (2 - 16);
    const int j = blockIdx.y * TS_BY + threadIdx.y;
    const int k = blockIdx.z;
    if (i < 0 || i > b.ib || j > b.jb) return;
    const long c = b.idx(i, j, k);
    b.ss[c] = b.p[c] / pow(b.w[c], b.gamma[c]);
}

__global__ __launch_bounds__(TS_BX* TS_BY) void k_entropy_level(const BlkView* __restrict__ tab, int nzb)
{
    const BlkView& b = tab[blockIdx.z / nzb + 1];
    const int i = blockIdx.x * TS_BX + threadIdx.x + (2 - 16);
    const int j = blockIdx.y * TS_BY + threadIdx.y;
    const int k = blockIdx.z % nzb;
    if (b.nx == 0 || i < 0 || i > b.ib || j > b.jb || k > b.kb) return;
    const long c = b.idx(i, j, k);
    b.ss[c] = b.p[c] / pow(b.w[c], b.gamma[c]);
}

void launch_entropy_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, hipStream_t s)
{
    LEVEL_SPLIT(nslots, maxnz + 4, launch_entropy_level(tab + s0_, n_, maxnx, maxny, maxnz, s));
    if (nslots <= 0) return;
    const int nzb = maxnz + 4;
    dim3 grd((maxnx + 4 + 14 + TS_BX - 1) / TS_BX, (maxny + 4 + TS_BY - 1) / TS_BY, nzb * nslots);
    hipLaunchKernelGGL(k_entropy_level, grd, dim3(TS_BX, TS_BY, 1), 0, s, tab, nzb);
}

void launch_entropy(const BlkView& b, hipStream_t s)
{
    dim3 blk(TS_BX, TS_BY, 1);
    dim3 grd((b.ib + 1 + 14 + TS_BX - 1) / TS_BX, (b.jb + 1 + TS_BY - 1) / TS_BY, b.kb + 1);
    hipLaunchKernelGGL(k_entropy, grd, blk, 0, s, b);
}

void launch_time_step_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const KParams& kp, hipStream_t s)
{
    LEVEL_SPLIT(nslots, maxnz + 4, launch_time_step_level(tab + s0_, n_, maxnx, maxny, maxnz, kp, s));
    if (nslots <= 0) return;
    dim3 blk(TS_BX, TS_BY, 1);
    const int nzb = maxnz + 4;
    dim3 grd((maxnx + 4 + 14 + TS_BX - 1) / TS_BX, (maxny + 4 + TS_BY - 1) / TS_BY, nzb * nslots);
    if (kp.viscous) {
        if (kp.doScaling)
            hipLaunchKernelGGL((k_time_step<true, true>), grd, blk, 0, s, tab, nzb, kp);
        else
            hipLaunchKernelGGL((k_time_step<true, false>), grd, blk, 0, s, tab, nzb, kp);
    } else {
        if (kp.doScaling)
            hipLaunchKernelGGL((k_time_step<false, true>), grd, blk, 0, s, tab, nzb, kp);
        else
            hipLaunchKernelGGL((k_time_step<false, false>), grd, blk, 0, s, tab, nzb, kp);
    }
}
