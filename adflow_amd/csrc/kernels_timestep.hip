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

    // i-direction: faces i-1 and i
    double sx = b.sI[c - 1] + b.sI[c];
    double sy = b.sI[c - 1 + nb] + b.sI[c + nb];
    double sz = b.sI[c - 1 + 2 * nb] + b.sI[c + 2 * nb];
    const double si2 = sx * sx + sy * sy + sz * sz;
    // grid velocity of a moving block: sum over the two faces (solverUtils.F90:147-181)
    double sFace = 0.0, sFaceJ = 0.0, sFaceK = 0.0;
    if (b.sFace) {      // uniform branch: no loads at all for blocks at rest
        const double* sF = b.sFace;
        sFace = sF[c - 1] + sF[c];
        sFaceJ = sF[c - b.ldi + nb] + sF[c + nb];
        sFaceK = sF[c - b.ldk + 2 * nb] + sF[c + 2 * nb];
    }
    double ri = 0.5 * (fabs(uux * sx + uuy * sy + uuz * sz - sFace) + kp.acousticScaleFactor * sqrt(cc2 * si2));

    sx = b.sJ[c - b.ldi] + b.sJ[c];
    sy = b.sJ[c - b.ldi + nb] + b.sJ[c + nb];
    sz = b.sJ[c - b.ldi + 2 * nb] + b.sJ[c + 2 * nb];
    const double sj2 = sx * sx + sy * sy + sz * sz;
    sFace = sFaceJ;
    double rj = 0.5 * (fabs(uux * sx + uuy * sy + uuz * sz - sFace) + kp.acousticScaleFactor * sqrt(cc2 * sj2));

    sx = b.sK[c - b.ldk] + b.sK[c];
    sy = b.sK[c - b.ldk + nb] + b.sK[c + nb];
    sz = b.sK[c - b.ldk + 2 * nb] + b.sK[c + 2 * nb];
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
