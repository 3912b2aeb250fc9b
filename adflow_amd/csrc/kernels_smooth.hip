// Smoother kernels: Runge-Kutta stage update, implicit residual averaging and
// the diagonalised ADI (D-ADI) factorisation.
//
// Reference semantics:
//   RungeKuttaSmoother / executeRkStage   src/solver/smoothers.F90:4-382
//   DADISmoother / executeDADIStep        src/solver/smoothers.F90:383-693
//   computedwDADI, tridiagsolve           src/solver/residuals.F90:1062-1783
//   residualAveraging                     src/solver/residuals.F90:1785-2082
//   computeEtotBlock                      src/utils/flowUtils.F90:551-672
//   computeLamViscosity                   src/utils/flowUtils.F90:1201-1300
//   saEddyViscosity                       src/turbulence/turbUtils.F90:657-720
//
// All pointwise work of a stage (dt scaling, conservative update with the
// density/pressure clipping, total energy, Sutherland viscosity, SA eddy
// viscosity) is ONE kernel.  Line solves along j and k map lanes to i
// (coalesced); lines along i are solved one line per lane.
// Roofline: HBM; no MFMA.
#include "internal.h"

#define SM_BX 64
#define SM_BY 4

__global__ __launch_bounds__(SM_BX* SM_BY) void k_rk_save(const BlkView* __restrict__ tab, int nzb)
{
    const BlkView& b = tab[blockIdx.z / nzb + 1];
    const int i = blockIdx.x * SM_BX + threadIdx.x + 2;
    const int j = blockIdx.y * SM_BY + threadIdx.y + 2;
    const int k = blockIdx.z % nzb + 2;
    if (i > b.il || j > b.jl || k > b.kl) return;
    const long c = b.idx(i, j, k);
#pragma unroll
    for (int l = 0; l < 5; ++l) b.wn[c + l * b.nbox] = b.w[c + l * b.nbox];
    b.pn[c] = b.p[c];
}

int g_max_grid_z = 65535;   // tuning "max_grid_z" (tests): the launchers split a level into slot ranges that fit gridDim.z

static dim3 level_grid(int nslots, int maxnx, int maxny, int maxnz)
{
    return dim3((maxnx + SM_BX - 1) / SM_BX, (maxny + SM_BY - 1) / SM_BY, maxnz * nslots);
}

// the pointwise smoother kernels cover every block of a level in one launch (blockIdx.z = slot * maxnz + plane)
void launch_rk_save_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, hipStream_t s)
{
    LEVEL_SPLIT(nslots, maxnz + 4, launch_rk_save_level(tab + s0_, n_, maxnx, maxny, maxnz, s));
    if (nslots <= 0) return;
    hipLaunchKernelGGL(k_rk_save, level_grid(nslots, maxnx, maxny, maxnz), dim3(SM_BX, SM_BY, 1), 0, s, tab, maxnz);
}

// computeEtotBlock on the owned cells (flowUtils.F90:551-672, cpConstant): the
// closing step of whalo2 when both p and rhoE were exchanged (haloExchange.F90:178-196)
__global__ __launch_bounds__(SM_BX* SM_BY) void k_etot_owned(BlkView b, double gammaConstant)
{
    const int i = blockIdx.x * SM_BX + threadIdx.x + 2;
    const int j = blockIdx.y * SM_BY + threadIdx.y + 2;
    const int k = blockIdx.z + 2;
    if (i > b.il || j > b.jl) return;
    const long c = b.idx(i, j, k);
    const long nb = b.nbox;
    const double ovgm1 = 1.0 / (gammaConstant - 1.0);
    const double u = b.w[c + nb], v = b.w[c + 2 * nb], w = b.w[c + 3 * nb];
    b.w[c + 4 * nb] = ovgm1 * b.p[c] + 0.5 * b.w[c] * (u * u + v * v + w * w);
}

// onlyIf: the pass runs only when *onlyIf != 0 (the matrix-free residual: k_set_w_closures_level has rewritten the energy already unless a
// pressure hit its floor)
// KCOL: a workgroup walks the planes of its 64 x 4 column (1 / nz of the workgroups: the launch that only finds the flag down costs
// the dispatch of a few hundred workgroups instead of tens of thousands)
template <bool KCOL>
__global__ __launch_bounds__(SM_BX* SM_BY) void k_etot_owned_level(const BlkView* __restrict__ tab, int nzb, double gammaConstant,
                                                                   const int* __restrict__ onlyIf)
{
    if (onlyIf && *onlyIf == 0) return;
    const BlkView& b = tab[(KCOL ? blockIdx.z : blockIdx.z / nzb) + 1];
    const int i = blockIdx.x * SM_BX + threadIdx.x + 2;
    const int j = blockIdx.y * SM_BY + threadIdx.y + 2;
    if (i > b.il || j > b.jl) return;
    const int k0 = KCOL ? 2 : (int)(blockIdx.z % nzb) + 2, k1 = KCOL ? b.kl : k0;
    const long nb = b.nbox;
    const double ovgm1 = 1.0 / (gammaConstant - 1.0);
    for (int k = k0; k <= k1 && k <= b.kl; ++k) {
        const long c = b.idx(i, j, k);
        const double u = b.w[c + nb], v = b.w[c + 2 * nb], w = b.w[c + 3 * nb];
        b.w[c + 4 * nb] = ovgm1 * b.p[c] + 0.5 * b.w[c] * (u * u + v * v + w * w);
    }
}

void launch_etot_owned_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, double gammaConstant, hipStream_t s,
                             const int* onlyIf)
{
    LEVEL_SPLIT(nslots, maxnz + 4, launch_etot_owned_level(tab + s0_, n_, maxnx, maxny, maxnz, gammaConstant, s, onlyIf));
    if (nslots <= 0) return;
    if (onlyIf) {
        // (the rare case: it runs only when a pressure hit its floor)
        const dim3 g((maxnx + SM_BX - 1) / SM_BX, (maxny + SM_BY - 1) / SM_BY, nslots);
        hipLaunchKernelGGL(k_etot_owned_level<true>, g, dim3(SM_BX, SM_BY, 1), 0, s, tab, maxnz, gammaConstant, onlyIf);
        return;
    }
    hipLaunchKernelGGL(k_etot_owned_level<false>, level_grid(nslots, maxnx, maxny, maxnz), dim3(SM_BX, SM_BY, 1), 0, s, tab, maxnz,
                       gammaConstant, onlyIf);
}

void launch_etot_owned(const BlkView& b, double gammaConstant, hipStream_t s)
{
    dim3 blk(SM_BX, SM_BY, 1);
    dim3 grd((b.nx + SM_BX - 1) / SM_BX, (b.ny + SM_BY - 1) / SM_BY, b.nz);
    hipLaunchKernelGGL(k_etot_owned, grd, blk, 0, s, b, gammaConstant);
}

// dw *= factor * dtl  [* vol]  (smoothers.F90:196-218 RK, :514-532 DADI)
__global__ __launch_bounds__(SM_BX* SM_BY) void k_scale_dw(const BlkView* __restrict__ tab, int nzb, double factor, int timesVol)
{
    const BlkView& b = tab[blockIdx.z / nzb + 1];
    const int i = blockIdx.x * SM_BX + threadIdx.x + 2;
    const int j = blockIdx.y * SM_BY + threadIdx.y + 2;
    const int k = blockIdx.z % nzb + 2;
    if (i > b.il || j > b.jl || k > b.kl) return;
    const long c = b.idx(i, j, k);
    double dt = factor * b.dtl[c];
    if (timesVol) dt *= b.vol[c];
#pragma unroll
    for (int l = 0; l < 5; ++l) b.dw[c + l * b.nbox] *= dt;
}

void launch_scale_dw_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, double factor, int timesVol, hipStream_t s)
{
    LEVEL_SPLIT(nslots, maxnz + 4, launch_scale_dw_level(tab + s0_, n_, maxnx, maxny, maxnz, factor, timesVol, s));
    if (nslots <= 0) return;
    hipLaunchKernelGGL(k_scale_dw, level_grid(nslots, maxnx, maxny, maxnz), dim3(SM_BX, SM_BY, 1), 0, s, tab, maxnz, factor, timesVol);
}

// sourceTerms_block (residuals.F90:348-425): body force and heat source of the cells of an actuator region.
// withBlank: applied to the finished residual of residual_block, i.e. times max(iblank, 0) as the sum there; without it
// the form blocketteRes applies after its core (blockette.F90:276-281).
__global__ __launch_bounds__(256) void k_source_terms(const BlkView* __restrict__ tab, const int* __restrict__ blk,
                                                      const long* __restrict__ off, int n, double fx, double fy, double fz, double qf,
                                                      int withBlank)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const BlkView& b = tab[blk[t]];
    const long c = off[t], nb = b.nbox;
    const double vol = b.vol[c];
    const double f1 = vol * fx, f2 = vol * fy, f3 = vol * fz, q = vol * qf;
    const double vx = b.w[c + nb], vy = b.w[c + 2 * nb], vz = b.w[c + 3 * nb];
    const double s = withBlank ? flg_blank(b.flags[c]) : 1.0;
    b.dw[c + nb] -= s * f1;
    b.dw[c + 2 * nb] -= s * f2;
    b.dw[c + 3 * nb] -= s * f3;
    b.dw[c + 4 * nb] -= s * (f1 * vx + f2 * vy + f3 * vz + q);
}

void launch_source_terms(const BlkView* tab, const int* blk, const long* off, int n, const double Ffact[3], double Qfact, int withBlank,
                         hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_source_terms, dim3((n + 255) / 256), dim3(256), 0, s, tab, blk, off, n, Ffact[0], Ffact[1], Ffact[2], Qfact,
                       withBlank);
}

// Low-speed preconditioner of residual_block (residuals.F90:172-331): dw <- B(w,p,gamma) * dw on the owned cells, with B the
// 5x5 product of the conservative->primitive jacobian and the low-Mach matrix A of that routine (K1, K2, M0 as there).
__global__ __launch_bounds__(SM_BX* SM_BY) void k_low_speed_precond(const BlkView* __restrict__ tab, int nzb, double uInf2)
{
    const BlkView& b = tab[blockIdx.z / nzb + 1];
    const int i = blockIdx.x * SM_BX + threadIdx.x + 2;
    const int j = blockIdx.y * SM_BY + threadIdx.y + 2;
    const int k = blockIdx.z % nzb + 2;
    if (i > b.il || j > b.jl || k > b.kl) return;
    const long c = b.idx(i, j, k), nb = b.nbox;
    constexpr double K1 = 1.05, K2 = 0.6, M0 = 0.2;
    const double rho = b.w[c], u = b.w[c + nb], v = b.w[c + 2 * nb], w = b.w[c + 3 * nb];
    const double gam = b.gamma[c], g1 = gam - 1.0;
    const double SoS = sqrt(gam * b.p[c] / rho), a2 = SoS * SoS, a4 = a2 * a2;
    const double q = u * u + v * v + w * w;
    const double resM = sqrt(q) / SoS, M2 = resM * resM;
    const double K3 = K1 * (1.0 + ((1.0 - K1 * M0 * M0) * M2) / (K1 * M0 * M0 * M0 * M0));
    const double betaMr2 = fmin(fmax(K3 * q, K2 * uInf2), a2);
    // rows of A: only columns 1, the diagonal and 5 are populated
    const double A1[5] = {betaMr2 * (1.0 / a4), 0.0, 0.0, 0.0, -betaMr2 / a4};
    const double A2[5] = {u / a2, rho, 0.0, 0.0, -u / a2};
    const double A3[5] = {v / a2, 0.0, rho, 0.0, -v / a2};
    const double A4[5] = {w / a2, 0.0, 0.0, rho, -w / a2};
    const double A5[5] = {1.0 / g1 + M2 / 2.0, rho * u, rho * v, rho * w, -M2 / 2.0};
    double d[5];
#pragma unroll
    for (int l = 0; l < 5; ++l) d[l] = b.dw[c + l * nb];
    const double h = g1 * q / 2.0, omg = 1.0 - gam;
    auto row = [&](const double* A) {
        const double B1 = A[0] * h + A[1] * (-u) / rho + A[2] * (-v) / rho + A[3] * (-w) / rho + A[4] * (h - a2);
        const double B2 = A[0] * omg * u + A[1] / rho + A[4] * omg * u;
        const double B3 = A[0] * omg * v + A[2] / rho + A[4] * omg * v;
        const double B4 = A[0] * omg * w + A[3] / rho + A[4] * omg * w;
        const double B5 = A[0] * g1 + A[4] * g1;
        return B1 * d[0] + B2 * d[1] + B3 * d[2] + B4 * d[3] + B5 * d[4];
    };
    b.dw[c] = row(A1);
    b.dw[c + nb] = row(A2);
    b.dw[c + 2 * nb] = row(A3);
    b.dw[c + 3 * nb] = row(A4);
    b.dw[c + 4 * nb] = row(A5);
}

void launch_low_speed_precond_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const KParams& kp, hipStream_t s)
{
    LEVEL_SPLIT(nslots, maxnz + 4, launch_low_speed_precond_level(tab + s0_, n_, maxnx, maxny, maxnz, kp, s));
    if (nslots <= 0) return;
    const double uInf2 = kp.wInf[1] * kp.wInf[1] + kp.wInf[2] * kp.wInf[2] + kp.wInf[3] * kp.wInf[3];
    hipLaunchKernelGGL(k_low_speed_precond, level_grid(nslots, maxnx, maxny, maxnz), dim3(SM_BX, SM_BY, 1), 0, s, tab, maxnz, uInf2);
}

// The update of one cell from its increment d (executeRkStage / executeDADIStep, smoothers.F90:292-380, 600-691): the pressure
// increment from the conservative ones, density and pressure floors, computeEtotBlock, Sutherland, SA eddy viscosity.
// (rho0 .. p0): the state the increment was linearised at; (rhoB .. pB): the state it is subtracted from.
__device__ __forceinline__ void stage_update_cell(const BlkView& b, const KParams& kp, long c, const double d[5], double rho0, double u0,
                                                  double v0, double w0, double e0, double p0, double rhoB, double uB, double vB, double wB,
                                                  double pB)
{
    const long nb = b.nbox;
    const double gm1 = kp.gammaConstant - 1.0;      // cpConstant, the only cp model of the path: gamma(i,j,k) = gammaConstant
    double ovr = rcp_nr(rho0);      // (reciprocals and the square root as v_rcp / v_rsq + one Newton step, internal.h)
    const double v2 = u0 * u0 + v0 * v0 + w0 * w0;
    const double dp = (ovr * p0 - gm1 * (ovr * e0 - v2)) * d[0] + gm1 * (d[4] - u0 * d[1] - v0 * d[2] - w0 * d[3]);
    const double ru = rhoB * uB - d[1], rv = rhoB * vB - d[2], rw = rhoB * wB - d[3];
    double rho = rhoB - d[0];
    rho = fmax(rho, 1.e-4 * kp.rhoInf);
    ovr = rcp_nr(rho);
    const double u = ovr * ru, v = ovr * rv, w = ovr * rw;
    double p = pB - dp;
    p = fmax(p, 1.e-4 * kp.pInfCorr);
    b.w[c] = rho;
    b.w[c + nb] = u;
    b.w[c + 2 * nb] = v;
    b.w[c + 3 * nb] = w;
    b.p[c] = p;
    // computeEtotBlock, cpConstant
    const double ovgm1 = 1.0 / (kp.gammaConstant - 1.0);
    b.w[c + 4 * nb] = ovgm1 * p + 0.5 * rho * (u * u + v * v + w * w);
    if (kp.viscous) {
        // computeLamViscosity (Sutherland)
        const double muSuth = kp.muSuthDim / kp.muRef, TSuth = kp.TSuthDim / kp.TRef, SSuth = kp.SSuthDim / kp.TRef;
        const double T = p * ovr * (1.0 / kp.RGas);
        const double tt = T * (1.0 / TSuth);
        const double rlv = muSuth * ((TSuth + SSuth) * rcp_nr(T + SSuth)) * (tt * fastsqrt(tt));
        b.rlv[c] = rlv;
        if (kp.eddyModel && kp.updateEddy) {
            const double cv13 = kp.sa_cv1 * kp.sa_cv1 * kp.sa_cv1;
            const double rnuSA = b.w[c + 5 * nb] * rho;
            const double chi = rnuSA * rcp_nr(rlv);
            const double chi3 = chi * chi * chi;
            b.rev[c] = chi3 * rcp_nr(chi3 + cv13) * rnuSA;
        }
    }
}

// State update of one stage.  FROM_WN: Runge-Kutta (new = stage-0 state - dw),
// otherwise D-ADI (new = current - dw).  scale != 0: dw is first multiplied by
// scale*dtl (fused k_scale_dw when no residual averaging sits in between).
template <bool FROM_WN>
__global__ __launch_bounds__(SM_BX* SM_BY) void k_stage_update(const BlkView* __restrict__ tab, int nzb, KParams kp, double scale)
{
    const BlkView& b = tab[blockIdx.z / nzb + 1];
    const int i = blockIdx.x * SM_BX + threadIdx.x + 2;
    const int j = blockIdx.y * SM_BY + threadIdx.y + 2;
    const int k = blockIdx.z % nzb + 2;
    if (i > b.il || j > b.jl || k > b.kl) return;
    const long c = b.idx(i, j, k);
    const long nb = b.nbox;
    double d[5];
#pragma unroll
    for (int l = 0; l < 5; ++l) d[l] = b.dw[c + l * nb];
    if (scale != 0.0) {
        const double dt = scale * b.dtl[c];
#pragma unroll
        for (int l = 0; l < 5; ++l) d[l] *= dt;
    }
    const double rho0 = b.w[c], u0 = b.w[c + nb], v0 = b.w[c + 2 * nb], w0 = b.w[c + 3 * nb], e0 = b.w[c + 4 * nb];
    const double p0 = b.p[c];
    double rhoB, uB, vB, wB, pB;
    if (FROM_WN) {
        rhoB = b.wn[c]; uB = b.wn[c + nb]; vB = b.wn[c + 2 * nb]; wB = b.wn[c + 3 * nb]; pB = b.pn[c];
    } else {
        rhoB = rho0; uB = u0; vB = v0; wB = w0; pB = p0;
    }
    stage_update_cell(b, kp, c, d, rho0, u0, v0, w0, e0, p0, rhoB, uB, vB, wB, pB);
}

void launch_stage_update_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const KParams& kp, double scale,
                               int fromWn, hipStream_t s)
{
    LEVEL_SPLIT(nslots, maxnz + 4, launch_stage_update_level(tab + s0_, n_, maxnx, maxny, maxnz, kp, scale, fromWn, s));
    if (nslots <= 0) return;
    const dim3 grd = level_grid(nslots, maxnx, maxny, maxnz), blk(SM_BX, SM_BY, 1);
    if (fromWn)
        hipLaunchKernelGGL((k_stage_update<true>), grd, blk, 0, s, tab, maxnz, kp, scale);
    else
        hipLaunchKernelGGL((k_stage_update<false>), grd, blk, 0, s, tab, maxnz, kp, scale);
}

// ---------------------------------------------------------------------------
// implicit residual averaging (residuals.F90:1856-2080): one line per lane.
// DIR 0: lines along i (lanes over j), DIR 1: along j (lanes over i),
// DIR 2: along k (lanes over i).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double ra_rfl(const BlkView& b, long c, double plim)
{
    const double p0 = b.p[c];
    double pa = b.p[c + 1], pb = b.p[c - 1];
    const double dpi = fabs(pa - 2.0 * p0 + pb) / (pa + 2.0 * p0 + pb + plim);
    pa = b.p[c + b.ldi]; pb = b.p[c - b.ldi];
    const double dpj = fabs(pa - 2.0 * p0 + pb) / (pa + 2.0 * p0 + pb + plim);
    pa = b.p[c + b.ldk]; pb = b.p[c - b.ldk];
    const double dpk = fabs(pa - 2.0 * p0 + pb) / (pa + 2.0 * p0 + pb + plim);
    return 1.0 / (1.0 + 2.0 * (dpi + dpj + dpk));
}

// Level-batched form: ONE launch per direction covers every block of the level and
// each of the 5 equations of a line is its own thread (the tridiagonal factor depends
// only on p, so the equations are independent): 5 x nBlocks x more lines in flight
// than one-line-per-lane-per-block, which is what hides the latency of the serial
// Thomas recurrences.  The pressure switch rfl is evaluated once per cell by a
// pointwise pre-pass (scratch 0) instead of once per cell, direction and sweep.
// scratch 1..5: the eliminated super-diagonal d of each equation's own solve.
__global__ __launch_bounds__(SM_BX* SM_BY) void k_ra_rfl(const BlkView* __restrict__ tab, KParams kp, int maxnz)
{
    const BlkView& b = tab[blockIdx.z / maxnz + 1];
    const int i = blockIdx.x * SM_BX + threadIdx.x + 2;
    const int j = blockIdx.y * SM_BY + threadIdx.y + 2;
    const int k = blockIdx.z % maxnz + 2;
    if (i > b.il || j > b.jl || k > b.kl) return;
    const long c = b.idx(i, j, k);
    b.scratch[c] = ra_rfl(b, c, 0.001 * kp.pInfCorr);
}

// A workgroup = 64 lines x the 5 equations (threadIdx.y): the tridiagonal factor depends only on the pressure switch, so
// every equation forms the same epz / t / d in registers, but only the wave of equation 0 STORES the eliminated
// super-diagonal d (scratch 1); after the workgroup barrier that follows the forward sweeps all five waves read it in
// their back substitution.  One d array instead of five: 184 instead of 320 B per cell and direction of HBM traffic with
// the parallelism of one thread per (line, equation) kept; rfl and the flags are shared through the CU's L1.
template <int DIR>
__global__ __launch_bounds__(64 * 5) void k_res_averaging(const BlkView* __restrict__ tab, KParams kp)
{
    const BlkView& b = tab[blockIdx.z + 1];
    const int l = threadIdx.y;
    // line coordinates (a fastest): DIR0 -> (j,k), DIR1 -> (i,k), DIR2 -> (i,j)
    const int a = blockIdx.x * 64 + threadIdx.x + 2;
    const int bb = blockIdx.y + 2;
    int n, amax, bmax;
    long c0, s;
    if (DIR == 0) { amax = b.jl; bmax = b.kl; n = b.nx; c0 = b.idx(2, a, bb); s = 1; }
    else if (DIR == 1) { amax = b.il; bmax = b.kl; n = b.ny; c0 = b.idx(a, 2, bb); s = b.ldi; }
    else { amax = b.il; bmax = b.jl; n = b.nz; c0 = b.idx(a, bb, 2); s = b.ldk; }
    if (b.nx == 0 || bb > bmax || n <= 1) return;        // uniform per workgroup
    const bool active = (a <= amax);
    const double rfl0 = 0.5 * kp.cfl / kp.cflLimit;
    const double* __restrict__ R = b.scratch;
    double* __restrict__ D = b.scratch + b.nbox;
    double* __restrict__ dw = b.dw + l * b.nbox;
    double prev = 0.0;                    // transformed dw(m-1)
    if (active) {
        // forward elimination (residuals.F90:1873-1895)
        double epzm = 0.0, dm = 0.0;          // epz(m-1), d(m-1) ; epz(1) = d(1) = 0
        double rflc = R[c0];
        for (int m = 0; m < n; ++m) {         // cell index 2+m along the line
            const long c = c0 + m * s;
            double epz = 0.0;
            double rfln = 0.0;
            if (m < n - 1) {                  // epz defined for 2..n (index il gets 0)
                rfln = R[c + s];
                const double r = rfl0 * (rflc + rfln);
                epz = 0.25 * kp.smoop * fmax(r * r - 1.0, 0.0) * flg_blank(b.flags[c]);
            }
            const double t = 1.0 / (1.0 + epz + epzm - epzm * dm);
            const double d = t * epz;
            if (l == 0) D[c] = d;
            const double v = t * (dw[c] + epzm * prev);
            dw[c] = v;
            prev = v;
            epzm = epz;
            dm = d;
            rflc = rfln;
        }
    }
    __syncthreads();       // d of the whole line set is in memory (written by the equation-0 wave of this workgroup)
    if (!active) return;
    // back substitution from index nx down to 2 (residuals.F90:1897-1905)
    for (int m = n - 2; m >= 0; --m) {
        const long c = c0 + m * s;
        const double v = dw[c] + D[c] * prev;
        dw[c] = v;
        prev = v;
    }
}

// Lines along i are contiguous in memory: with lanes over j a wave would touch 64
// different rows per access.  Here the 64 lines of a workgroup are processed in
// chunks of RA_CH cells that travel through an LDS tile: global loads / stores move
// 64-byte row segments (8 lines x 8 cells per wave access), the serial recurrence
// of line `lane` runs on the tile.  Same arithmetic as k_res_averaging<0>.
#define RA_SH 3
#define RA_CH (1 << RA_SH)
#define RA_LD (RA_CH + 1)
#define RA_LPA (64 >> RA_SH)     // lines moved per wave access

// The tile transposition runs through registers and ONE LDS tile per workgroup (one wave): the coalesced global loads of
// all arrays of a chunk are in flight together, and 4.6 KB of LDS per wave leave the occupancy to the registers
// (three tiles = 18.5 KB held it at 8 waves per CU; the kernel is bound by load latency).
// PASS 0: forward elimination (the eliminated diagonal d is the same for the five equations: only the equation-0 workgroups
// store it, scratch 1); PASS 1: back substitution in a second launch, whose boundary orders it after every d store.
// scaleDtl != 0 (PASS 0): the update enters the first solve as scaleDtl dtl dw, the scaling of the Runge-Kutta stage
// (smoothers.F90:202-230) that would otherwise be a pointwise pass over dw of its own
template <int PASS>
__global__ __launch_bounds__(64) void k_res_averaging_i(const BlkView* __restrict__ tab, KParams kp, double scaleDtl)
{
    __shared__ double tile[64 * RA_LD];
    const BlkView& b = tab[blockIdx.z + 1];
    const int l = blockIdx.x % 5;
    const int lane = threadIdx.x;
    const int j0 = (blockIdx.x / 5) * 64 + 2;
    const int k = blockIdx.y + 2;
    const int n = b.nx;
    if (j0 > b.jl || k > b.kl || n <= 1) return;
    const double rfl0 = 0.5 * kp.cfl / kp.cflLimit;
    const double* __restrict__ R = b.scratch;
    double* __restrict__ D = b.scratch + b.nbox;
    double* __restrict__ dw = b.dw + l * b.nbox;
    const int sub = lane >> RA_SH, col = lane & (RA_CH - 1);   // tile transfer role: RA_LPA lines x RA_CH cells per access
    const bool lineOk = (j0 + lane <= b.jl);
    const int nch = (n + RA_CH - 1) / RA_CH;
    // raw (coalesced order) -> the RA_CH cells of the lane's own line
    auto to_line = [&](const double raw[RA_CH], double v[RA_CH]) {
#pragma unroll
        for (int q = 0; q < RA_CH; ++q) tile[(RA_LPA * q + sub) * RA_LD + col] = raw[q];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < RA_CH; ++m) v[m] = tile[lane * RA_LD + m];
        __syncthreads();
    };
    auto to_global = [&](double* __restrict__ arr, int i0, const double v[RA_CH]) {
#pragma unroll
        for (int m = 0; m < RA_CH; ++m) tile[lane * RA_LD + m] = v[m];
        __syncthreads();
        const int i = i0 + col;
#pragma unroll
        for (int q = 0; q < RA_CH; ++q) {
            const int r = RA_LPA * q + sub;
            if (j0 + r <= b.jl && i <= b.il) arr[b.idx(i, j0 + r, k)] = tile[r * RA_LD + col];
        }
        __syncthreads();
    };

    double epzm = 0.0, dm = 0.0, prev = 0.0;
    double rflc = (PASS == 0 && lineOk) ? R[b.idx(2, j0 + lane, k)] : 0.0;
    for (int ch = 0; PASS == 0 && ch < nch; ++ch) {
        const int i0 = 2 + ch * RA_CH;
        const int i = i0 + col;
        double rv[RA_CH], rr[RA_CH], rf[RA_CH];
#pragma unroll
        for (int q = 0; q < RA_CH; ++q) {
            const int r = RA_LPA * q + sub;
            rv[q] = 0.0; rr[q] = 0.0; rf[q] = 0.0;
            if (j0 + r <= b.jl && i <= b.il) {
                const long c = b.idx(i, j0 + r, k);
                rv[q] = dw[c]; rr[q] = R[c + 1]; rf[q] = flg_blank(b.flags[c]);
                if (scaleDtl != 0.0) rv[q] *= scaleDtl * b.dtl[c];
            }
        }
        double tv[RA_CH], tr[RA_CH], tb[RA_CH];
        to_line(rv, tv); to_line(rr, tr); to_line(rf, tb);
        const int mEnd = (n - ch * RA_CH < RA_CH) ? n - ch * RA_CH : RA_CH;
#pragma unroll
        for (int m = 0; m < RA_CH; ++m) {
            if (m < mEnd) {
                double epz = 0.0, rfln = 0.0;
                if (ch * RA_CH + m < n - 1) {
                    rfln = tr[m];
                    const double r = rfl0 * (rflc + rfln);
                    epz = 0.25 * kp.smoop * fmax(r * r - 1.0, 0.0) * tb[m];
                }
                const double t = 1.0 / (1.0 + epz + epzm - epzm * dm);
                const double d = t * epz;
                tr[m] = d;
                const double v = t * (tv[m] + epzm * prev);
                tv[m] = v;
                prev = v;
                epzm = epz;
                dm = d;
                rflc = rfln;
            }
        }
        to_global(dw, i0, tv);
        if (l == 0) to_global(D, i0, tr);        // uniform per workgroup (one wave)
    }
    if (PASS == 0) return;
    // back substitution: cells n-2 .. 0, chunks right to left
    for (int ch = nch - 1; ch >= 0; --ch) {
        const int i0 = 2 + ch * RA_CH;
        const int i = i0 + col;
        double rv[RA_CH], rd[RA_CH];
#pragma unroll
        for (int q = 0; q < RA_CH; ++q) {
            const int r = RA_LPA * q + sub;
            rv[q] = 0.0; rd[q] = 0.0;
            if (j0 + r <= b.jl && i <= b.il) {
                const long c = b.idx(i, j0 + r, k);
                rv[q] = dw[c]; rd[q] = D[c];
            }
        }
        double tv[RA_CH], tr[RA_CH];
        to_line(rv, tv); to_line(rd, tr);
        if (ch == nch - 1) {
            // the last cell of the line keeps its forward value: the start of the substitution
            const int mLast = n - 1 - ch * RA_CH;
#pragma unroll
            for (int m = 0; m < RA_CH; ++m)
                if (m == mLast) prev = tv[m];
        }
        int mTop = n - 2 - ch * RA_CH;              // last cell that is updated
        if (mTop > RA_CH - 1) mTop = RA_CH - 1;
#pragma unroll
        for (int m = RA_CH - 1; m >= 0; --m) {
            if (m <= mTop) {
                const double v = tv[m] + tr[m] * prev;
                tv[m] = v;
                prev = v;
            }
        }
        to_global(dw, i0, tv);
    }
}

// ---------------------------------------------------------------------------
// Residual averaging along i with the lines RESIDENT in LDS (round 3).  The two-pass kernels move every dw value four times per
// direction (forward: read + write, back substitution: read + write) plus the eliminated super-diagonal, and along i -- lines
// contiguous in memory, lanes over j -- every access goes through a tile transposition: 1.05 + 0.49 ms on 8 x 128^3 against 0.66 ms
// for j or k.  Here a workgroup loads a bundle of NL whole i lines once (coalesced), keeps the five equations, the coefficient epz
// and the eliminated super-diagonal d of every cell in LDS (7 doubles per cell), runs the Thomas recurrences with one thread per
// (line, equation) -- the factor depends only on the pressure switch, every equation thread forms it itself -- and stores the bundle
// once: 0.79 ms.  (Measured for j / k as well, bundles of 16 row segments: 0.87 ms against 0.66 ms -- 80 recurrence threads per CU
// instead of thousands; LDS holds no more lines.  They keep the two-pass kernels.)
// Same arithmetic as k_res_averaging_i (residuals.F90:1856-2080) up to rounding.
// NL, P (padded line length, odd): chosen by the launcher from the longest line of the level so that 7 NL P doubles fit the buffer.
// ---------------------------------------------------------------------------
#define RL_C 32             // positions per chunk of the recurrences
#define RL_BUF 9216         // doubles: 72 KB, two workgroups per CU
__global__ __launch_bounds__(256) void k_ra_line_i(const BlkView* __restrict__ tab, KParams kp, double scaleDtl, int NL, int P)
{
    __shared__ double buf[RL_BUF];
    const BlkView& b = tab[blockIdx.z + 1];
    const int tid = threadIdx.x;
    const int bb = blockIdx.y + 2;
    const int l0 = blockIdx.x * NL;                 // first line of the bundle (0-based)
    const int n = b.nx, nlines = b.ny;
    if (b.nx == 0 || bb > b.kl || n <= 1 || l0 >= nlines) return;          // uniform per workgroup
    const long c0 = b.idx(2, 2 + l0, bb), sl = b.ldi;                      // cell (line l, position m) = c0 + l sl + m
    const int nl = (nlines - l0 < NL) ? nlines - l0 : NL;
    const int nC = (n + RL_C - 1) & ~(RL_C - 1);    // the recurrences run in chunks of RL_C positions: nC <= P - 1, the tail holds zeros
    const long nb = b.nbox;
    double* __restrict__ X = buf;                   // [equation][line][m]
    double* __restrict__ E = buf + 5 * NL * P;      // epz [line][m]
    double* __restrict__ D = E + NL * P;            // eliminated super-diagonal [line][m]
    const double rfl0 = 0.5 * kp.cfl / kp.cflLimit;
    const double* __restrict__ R = b.scratch;
    // ---- the bundle -> LDS: a wave covers 64 cells of one line.  Four positions per thread are requested before the first is
    //      stored (their latencies overlap)
    const int lA = tid >> 6, lS = 4, mA = tid & 63, mS = 64;
    for (int l = lA; l < nl; l += lS)
        for (int m0 = mA; m0 < nC; m0 += 4 * mS) {
            double v[4][5], ep[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int m = m0 + u * mS;
                ep[u] = 0.0;
#pragma unroll
                for (int q = 0; q < 5; ++q) v[u][q] = 0.0;
                if (m < n) {
                    const long c = c0 + l * sl + m;
                    const double sc = (scaleDtl != 0.0) ? scaleDtl * b.dtl[c] : 1.0;
#pragma unroll
                    for (int q = 0; q < 5; ++q) v[u][q] = b.dw[c + q * nb] * sc;
                    if (m < n - 1) {
                        const double r = rfl0 * (R[c] + R[c + 1]);
                        ep[u] = 0.25 * kp.smoop * fmax(r * r - 1.0, 0.0) * flg_blank(b.flags[c]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int m = m0 + u * mS;
                if (m < nC) {
#pragma unroll
                    for (int q = 0; q < 5; ++q) X[(q * NL + l) * P + m] = v[u][q];
                    E[l * P + m] = ep[u];
                }
            }
        }
    __syncthreads();
    // ---- Thomas recurrences: one thread per (line, equation), the lines spread over the four waves (each wave pays the LDS round trip
    //      of a chunk on its own SIMD at the same time as the others).  Chunks of RL_C positions: an LDS round trip costs several
    //      hundred cycles here, so a line is a few long chunks, all of a chunk requested before its chain starts.
    //      Forward elimination without a division in the dependent chain: with q(m) the product of the pivots up to m,
    //        q(m) = (1 + epz(m) + epz(m-1)) q(m-1) - epz(m-1)^2 q(m-2),   u(m) = q(m-1) x(m) + epz(m-1) u(m-1)   (u = q v)
    //      are linear recurrences; the eliminated super-diagonal d(m) = epz(m) q(m-1) / q(m) and v(m) = u(m) / q(m) follow outside
    //      the chain.  q grows with the line: at every chunk q and u are scaled by the power of two of q (exact).
    {
        const int lpw = (NL + 3) >> 2, lane = tid & 63;
        const int l = (tid >> 6) * lpw + lane / 5, q = lane % 5;
        if (lane < 5 * lpw && l < nl) {
            double* __restrict__ x = X + (q * NL + l) * P;
            const double* __restrict__ e = E + l * P;
            double* __restrict__ d = D + l * P;
            double qm1 = 1.0, qm2 = 0.0, um1 = 0.0, epzm = 0.0;
            for (int m0 = 0; m0 < nC; m0 += RL_C) {
                double ec[RL_C], xc[RL_C];
#pragma unroll
                for (int u = 0; u < RL_C; ++u) { ec[u] = e[m0 + u]; xc[u] = x[m0 + u]; }
                {
                    const int ks = -exponent_of(qm1);
                    qm1 = __builtin_ldexp(qm1, ks); qm2 = __builtin_ldexp(qm2, ks); um1 = __builtin_ldexp(um1, ks);
                }
#pragma unroll
                for (int u = 0; u < RL_C; ++u) {
                    const double epz = ec[u];
                    const double qq = (1.0 + epz + epzm) * qm1 - (epzm * epzm) * qm2;
                    const double uu = qm1 * xc[u] + epzm * um1;
                    const double r = rcp_nr(qq);
                    d[m0 + u] = (epz * qm1) * r;    // the same value from the five equation threads of the line
                    x[m0 + u] = uu * r;
                    qm2 = qm1; qm1 = qq; um1 = uu; epzm = epz;
                }
            }
            // back substitution from position n-1 (its forward value) down to 0: v(m) += d(m) v(m+1)
            double prev = x[n - 1];
            int mTop = n - 2;
            for (; mTop >= 0 && ((mTop + 1) & (RL_C - 1)); --mTop) {          // down to a chunk boundary
                const double vv = x[mTop] + d[mTop] * prev;
                x[mTop] = vv;
                prev = vv;
            }
            for (int m0 = mTop - (RL_C - 1); m0 >= 0; m0 -= RL_C) {
                double dc[RL_C], xc[RL_C];
#pragma unroll
                for (int u = 0; u < RL_C; ++u) { dc[u] = d[m0 + u]; xc[u] = x[m0 + u]; }
#pragma unroll
                for (int u = RL_C - 1; u >= 0; --u) {
                    const double vv = xc[u] + dc[u] * prev;
                    x[m0 + u] = vv;
                    prev = vv;
                }
            }
        }
    }
    __syncthreads();
    for (int l = lA; l < nl; l += lS)
        for (int m = mA; m < n; m += mS) {
            const long c = c0 + l * sl + m;
#pragma unroll
            for (int q = 0; q < 5; ++q) b.dw[c + q * nb] = X[(q * NL + l) * P + m];
        }
}

// Round 4: residual averaging along i by parallel cyclic reduction along the lanes (the scheme of k_dadi_i_pcr below): a workgroup of
// NW wavefronts holds one i line, lane m the row  -epz(m-1) x(m-1) + (1 + epz(m) + epz(m-1)) x(m) - epz(m) x(m+1) = dw(m)  of the five
// equations (one coefficient set), normalised by its diagonal; ceil(log2 nx) steps through LDS, all lanes busy, coalesced row
// accesses and no transposition: one read and one write of dw.  k_ra_line_i keeps the lines of more than 256 cells.
#define RP_JL 8
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_ra_i_pcr(const BlkView* __restrict__ tab, KParams kp, double scaleDtl)
{
    constexpr int T = 64 * NW;
    __shared__ double P[2 * 7 * T];
    const BlkView& b = tab[blockIdx.z + 1];
    const int t = threadIdx.x, n = b.nx;
    const int k = blockIdx.y + 2;
    const int j0 = blockIdx.x * RP_JL + 2;
    if (b.nx == 0 || k > b.kl || j0 > b.jl || n <= 1 || n > T) return;        // uniform per workgroup
    const bool act = t < n;
    const int tc = act ? t : n - 1;
    const long nb = b.nbox;
    const double rfl0 = 0.5 * kp.cfl / kp.cflLimit;
    const double* __restrict__ R = b.scratch;
    for (int jl_ = 0; jl_ < RP_JL; ++jl_) {
        const int j = j0 + jl_;
        if (j > b.jl) break;                                         // uniform
        const long c = b.idx(2 + tc, j, k);
        const double sc = (scaleDtl != 0.0) ? scaleDtl * b.dtl[c] : 1.0;
        double d[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) d[q] = act ? b.dw[c + q * nb] * sc : 0.0;
        // epz(m) lives between the cells m and m+1 (0 beyond the line ends)
        const double r0 = R[c], rp = R[c + 1], rm = R[c - 1];
        double epz = 0.0, epzm = 0.0;
        if (act && t < n - 1) { const double r = rfl0 * (r0 + rp); epz = 0.25 * kp.smoop * fmax(r * r - 1.0, 0.0) * flg_blank(b.flags[c]); }
        if (act && t > 0) { const double r = rfl0 * (rm + r0); epzm = 0.25 * kp.smoop * fmax(r * r - 1.0, 0.0) * flg_blank(b.flags[c - 1]); }
        const double inv = rcp_nr(1.0 + epz + epzm);
        double a = -epzm * inv, cc = -epz * inv;
#pragma unroll
        for (int q = 0; q < 5; ++q) d[q] *= inv;
        int cur_ = 0;
        for (int st = 1; st < n; st <<= 1) {
            double* __restrict__ Q = P + cur_ * 7 * T;
            Q[t] = a; Q[T + t] = cc;
#pragma unroll
            for (int q = 0; q < 5; ++q) Q[(2 + q) * T + t] = d[q];
            __syncthreads();
            const bool lo = t >= st, hi = t + st < n;
            const int im = lo ? t - st : t, ip = hi ? t + st : t;
            const double am = lo ? Q[im] : 0.0, cm = lo ? Q[T + im] : 0.0;
            const double ap = hi ? Q[ip] : 0.0, cp = hi ? Q[T + ip] : 0.0;
            const double al = -a, ga = -cc;
            const double iv = rcp_nr(1.0 + al * cm + ga * ap);
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const double dm = lo ? Q[(2 + q) * T + im] : 0.0, dp = hi ? Q[(2 + q) * T + ip] : 0.0;
                d[q] = (d[q] + al * dm + ga * dp) * iv;
            }
            a = al * am * iv;
            cc = ga * cp * iv;
            cur_ ^= 1;
        }
        if (act) {
#pragma unroll
            for (int q = 0; q < 5; ++q) b.dw[c + q * nb] = d[q];
        }
        __syncthreads();
    }
}

template <int NW>
static void launch_ra_i_pcr(const BlkView* tab, int nslots, int ny, int nz, const KParams& kp, double scaleDtl, hipStream_t s)
{
    hipLaunchKernelGGL((k_ra_i_pcr<NW>), dim3((ny + RP_JL - 1) / RP_JL, nz, nslots), dim3(64 * NW, 1, 1), 0, s, tab, kp, scaleDtl);
}

int g_ra_pcr = 1;        // tuning "ra_pcr": residual averaging along i by cyclic reduction (0: LDS-resident lines / two-pass kernels)

// scaleDtl != 0: only for levels whose blocks all have more than one cell in i (the scaling rides on the i sweep)
void launch_res_averaging_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const KParams& kp, hipStream_t s,
                                double scaleDtl)
{
    LEVEL_SPLIT(nslots, maxnz + 4, launch_res_averaging_level(tab + s0_, n_, maxnx, maxny, maxnz, kp, s, scaleDtl));
    if (nslots <= 0) return;
    hipLaunchKernelGGL(k_ra_rfl, dim3((maxnx + SM_BX - 1) / SM_BX, (maxny + SM_BY - 1) / SM_BY, maxnz * nslots),
                       dim3(SM_BX, SM_BY, 1), 0, s, tab, kp, maxnz);
    const dim3 blk(64, 1, 1);
    const dim3 blk5(64, 5, 1);        // 64 lines x 5 equations
    if (maxnx > 1 && maxnx <= 256 && g_ra_pcr) {
        if (maxnx <= 64) launch_ra_i_pcr<1>(tab, nslots, maxny, maxnz, kp, scaleDtl, s);
        else if (maxnx <= 128) launch_ra_i_pcr<2>(tab, nslots, maxny, maxnz, kp, scaleDtl, s);
        else if (maxnx <= 192) launch_ra_i_pcr<3>(tab, nslots, maxny, maxnz, kp, scaleDtl, s);
        else launch_ra_i_pcr<4>(tab, nslots, maxny, maxnz, kp, scaleDtl, s);
    } else if (maxnx > 1) {
        // i lines resident in LDS while at least two of them fit the buffer; longer lines: the two-pass kernels
        const int P = ((maxnx + RL_C - 1) & ~(RL_C - 1)) + 1;          // whole chunks, odd (LDS banks)
        const int fit = RL_BUF / (7 * P), NL = fit < 16 ? fit : 16;
        if (NL >= 2)
            hipLaunchKernelGGL(k_ra_line_i, dim3((maxny + NL - 1) / NL, maxnz, nslots), dim3(256), 0, s, tab, kp, scaleDtl, NL, P);
        else {
            hipLaunchKernelGGL((k_res_averaging_i<0>), dim3(5 * ((maxny + 63) / 64), maxnz, nslots), blk, 0, s, tab, kp, scaleDtl);
            hipLaunchKernelGGL((k_res_averaging_i<1>), dim3(5 * ((maxny + 63) / 64), maxnz, nslots), blk, 0, s, tab, kp, 0.0);
        }
    }
    if (maxny > 1) hipLaunchKernelGGL((k_res_averaging<1>), dim3((maxnx + 63) / 64, maxnz, nslots), blk5, 0, s, tab, kp);
    if (maxnz > 1) hipLaunchKernelGGL((k_res_averaging<2>), dim3((maxnx + 63) / 64, maxny, nslots), blk5, 0, s, tab, kp);
}

// ---------------------------------------------------------------------------
// D-ADI (residuals.F90:1062-1748)
// ---------------------------------------------------------------------------
struct DadiCell {          // per-cell coefficients of one direction
    double dP[3], dM[3];   // diagPlus / diagMinus for eigenvalue groups {1,2,3}, 4, 5
    double vt1, vt3;       // viscTerm1 (face m), viscTerm3 (face m-1)
    double ddt;            // dual_dt * max(iblank,0)
};

// DIR: 0 = i, 1 = j, 2 = k.  Coefficients of cell c (residuals.F90:1349-1376 for j), in two halves so that a marching thread can
// REQUEST the values of a cell one step before it forms the coefficients (round 4: the sweeps waited for these loads in every step)
struct DadiRaw {
    double vol, volP, volM, rho, u, v, w, p, n0[3], nm[3], mk[3], rlv[3], rev[3], dtl, qs;
    int flag;
};

template <int DIR>
__device__ __forceinline__ void dadi_load(const BlkView& b, const KParams& kp, long c, long s, const double* __restrict__ sN, DadiRaw& r)
{
    const long nb = b.nbox;
    r.vol = b.vol[c]; r.rho = b.w[c];
    r.u = b.w[c + nb]; r.v = b.w[c + 2 * nb]; r.w = b.w[c + 3 * nb];
    r.p = b.p[c];
#pragma unroll
    for (int q = 0; q < 3; ++q) { r.n0[q] = sN[c + q * nb]; r.nm[q] = sN[c - s + q * nb]; }
    // grid velocity of a moving block (residuals.F90:1192-1196)
    r.qs = 0.0;
    if (b.sFace) r.qs = b.sFace[c - s + DIR * nb] + b.sFace[c + DIR * nb];     // uniform branch
    if (DIR == 2) {
        // metric used in eps2: the k-direction mixes sK(k) with sJ(k-1) in the reference (residuals.F90:1625-1627): reproduced
#pragma unroll
        for (int q = 0; q < 3; ++q) r.mk[q] = b.sJ[c - s + q * nb];
    }
    if (kp.viscous) {
        r.rlv[0] = b.rlv[c - s]; r.rlv[1] = b.rlv[c]; r.rlv[2] = b.rlv[c + s];
        if (kp.eddyModel) { r.rev[0] = b.rev[c - s]; r.rev[1] = b.rev[c]; r.rev[2] = b.rev[c + s]; }
        r.volP = b.vol[c + s]; r.volM = b.vol[c - s];
    }
    r.dtl = b.dtl[c];
    r.flag = b.flags[c];
}

template <int DIR>
__device__ __forceinline__ void dadi_coef(const KParams& kp, const DadiRaw& r, DadiCell& o)
{
    // divisions and square roots as v_rcp / v_rsq + one Newton step (internal.h: 2e-15 relative): with one wavefront per SIMD the
    // sweeps are bound by the ~1200 instructions of a step, two thirds of them the compiler's IEEE division / sqrt sequences
    const double vol = r.vol, rho = r.rho, u = r.u, v = r.v, w = r.w;
    const double ovol = rcp_nr(vol), orho = rcp_nr(rho);
    const double volhalf = 0.5 * ovol;
    // mean normal of the two faces (velocity part)
    const double r1 = volhalf * (r.n0[0] + r.nm[0]);
    const double r2 = volhalf * (r.n0[1] + r.nm[1]);
    const double r3 = volhalf * (r.n0[2] + r.nm[2]);
    const double qs = r.qs * volhalf;
    const double qq = r1 * u + r2 * v + r3 * w - qs;
    const double cijk = fastsqrt(kp.gammaConstant * r.p * orho);      // (calorically perfect gas: gamma(i,j,k) = gammaConstant, as the marching kernels)
    const double cc = cijk * fastsqrt(r1 * r1 + r2 * r2 + r3 * r3);
    double m1 = r1, m2 = r2, m3 = r3;
    if (DIR == 2) {
        m1 = volhalf * (r.n0[0] + r.mk[0]);
        m2 = volhalf * (r.n0[1] + r.mk[1]);
        m3 = volhalf * (r.n0[2] + r.mk[2]);
    }
    const double epsval = 0.08, fac = 1.05;
    const double cInf2 = kp.gammaInf * kp.pInf / kp.rhoInf;
    const double eps2 = epsval * epsval * cInf2 * (m1 * m1 + m2 * m2 + m3 * m3);
    const double s0 = fac * fastsqrt(qq * qq + eps2), s1 = fac * fastsqrt((qq + cc) * (qq + cc) + eps2),
                 s2 = fac * fastsqrt((qq - cc) * (qq - cc) + eps2);
    o.dP[0] = 0.5 * (qq + s0);
    o.dP[1] = 0.5 * (qq + cc + s1);
    o.dP[2] = 0.5 * (qq - cc + s2);
    o.dM[0] = 0.5 * (qq - s0);
    o.dM[1] = 0.5 * (qq + cc - s1);
    o.dM[2] = 0.5 * (qq - cc - s2);
    // viscous terms: metterm(face) = |S|^2 * mut / (vol_m + vol_m+1)
    double mtP = 0.0, mtM = 0.0;
    if (kp.viscous) {
        double mutP = r.rlv[1] + r.rlv[2], mutM = r.rlv[0] + r.rlv[1];
        if (kp.eddyModel) {
            mutP += r.rev[1] + r.rev[2];
            mutM += r.rev[0] + r.rev[1];
        }
        const double sp = r.n0[0] * r.n0[0] + r.n0[1] * r.n0[1] + r.n0[2] * r.n0[2];
        const double sm = r.nm[0] * r.nm[0] + r.nm[1] * r.nm[1] + r.nm[2] * r.nm[2];
        mtP = sp * mutP * rcp_nr(vol + r.volP);
        mtM = sm * mutM * rcp_nr(r.volM + vol);
    }
    const double ovr = ovol * orho;
    o.vt1 = mtP * ovr;
    o.vt3 = mtM * ovr;
    o.ddt = kp.cfl * r.dtl * vol * flg_blank((uint8_t)r.flag);
}

template <int DIR>
__device__ __forceinline__ void dadi_cell(const BlkView& b, const KParams& kp, long c, long s, const double* __restrict__ sN,
                                          DadiCell& o)
{
    DadiRaw r;
    dadi_load<DIR>(b, kp, c, s, sN, r);
    dadi_coef<DIR>(kp, r, o);
}

// T_eta^-1 applied to the physical update (residuals.F90:1276-1331)
__device__ __forceinline__ void dadi_pre_j(const BlkView& b, long c, double d[5], double gam)
{
    const long nb = b.nbox;
    const double rho = b.w[c], uvel = b.w[c + nb], vvel = b.w[c + 2 * nb], wvel = b.w[c + 3 * nb];
    const double gm1 = gam - 1.0;
    const double cijk = sqrt(gam * b.p[c] / rho);
    const double c2inv = 1.0 / (cijk * cijk);
    const double xfact = 2.0 * cijk;
    const double alphinv = sqrt(2.0) * cijk / rho;
    const double uvw = 0.5 * (uvel * uvel + vvel * vvel + wvel * wvel);
    const long s = b.ldi;
    double rj1 = 0.5 * (b.sJ[c] + b.sJ[c - s]), rj2 = 0.5 * (b.sJ[c + nb] + b.sJ[c - s + nb]),
           rj3 = 0.5 * (b.sJ[c + 2 * nb] + b.sJ[c - s + 2 * nb]);
    const double rj = sqrt(rj1 * rj1 + rj2 * rj2 + rj3 * rj3);
    const double uu = uvel * rj1 + vvel * rj2 + wvel * rj3;
    rj1 /= rj; rj2 /= rj; rj3 /= rj;
    const double dw1 = d[0], dw2 = d[1], dw3 = d[2], dw4 = d[3], dw5 = d[4];
    double a1 = dw2 * uvel + dw3 * vvel + dw4 * wvel - dw5;
    a1 = a1 * gm1 * c2inv + dw1 * (1.0 - uvw * gm1 * c2inv);
    const double a2 = (rj2 * wvel - rj3 * vvel) * dw1 + rj3 * dw3 - rj2 * dw4;
    const double a3 = (rj3 * uvel - rj1 * wvel) * dw1 + rj1 * dw4 - rj3 * dw2;
    const double a4 = (rj1 * vvel - rj2 * uvel) * dw1 + rj2 * dw2 - rj1 * dw3;
    double a5 = uvw * dw1 - uvel * dw2 - vvel * dw3 - wvel * dw4 + dw5;
    a5 = a5 * gm1 * c2inv;
    const double a6 = uu * dw1 / rj - rj1 * dw2 - rj2 * dw3 - rj3 * dw4;
    d[0] = a1 * rj1 + a2 / rho;
    d[1] = a1 * rj2 + a3 / rho;
    d[2] = a1 * rj3 + a4 / rho;
    d[3] = (0.5 * a5 - a6 / xfact) * alphinv;
    d[4] = (0.5 * a5 + a6 / xfact) * alphinv;
}

// T_A^-1 T_B between two directions given their unit mean normals
// (residuals.F90:1404-1446 with (ri,rj); :1540-1583 with (rk,ri) in swapped roles)
__device__ __forceinline__ void dadi_rotate(const double a1, const double a2, const double a3, const double a4, double d[5])
{
    const double sqrt2inv = 1.0 / sqrt(2.0);
    const double dw1 = d[0], dw2 = d[1], dw3 = d[2], dw4 = d[3], dw5 = d[4];
    const double a5 = (dw4 - dw5) * sqrt2inv;
    const double a6 = (dw4 + dw5) * 0.5;
    const double a7 = (a3 * dw1 + a4 * dw2 - a2 * dw3 - a5 * a1) * sqrt2inv;
    d[0] = a1 * dw1 + a2 * dw2 + a4 * dw3 + a5 * a3;
    d[1] = -a2 * dw1 + a1 * dw2 - a3 * dw3 + a5 * a4;
    d[2] = -a4 * dw1 + a3 * dw2 + a1 * dw3 - a5 * a2;
    d[3] = -a7 + a6;
    d[4] = a7 + a6;
}

__device__ __forceinline__ void unit_mean_normal(const double* __restrict__ sN, long c, long s, long nb, double r[3])
{
    r[0] = 0.5 * (sN[c] + sN[c - s]);
    r[1] = 0.5 * (sN[c + nb] + sN[c - s + nb]);
    r[2] = 0.5 * (sN[c + 2 * nb] + sN[c - s + 2 * nb]);
    const double orr = rsq_nr(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    r[0] *= orr; r[1] *= orr; r[2] *= orr;
}

// after the j-solve: T_xi^-1 T_eta (residuals.F90:1404-1446)
__device__ __forceinline__ void dadi_post_j(const BlkView& b, long c, double d[5])
{
    double ri[3], rj[3];
    unit_mean_normal(b.sI, c, 1, b.nbox, ri);
    unit_mean_normal(b.sJ, c, b.ldi, b.nbox, rj);
    const double a1 = ri[0] * rj[0] + ri[1] * rj[1] + ri[2] * rj[2];
    const double a2 = ri[0] * rj[1] - rj[0] * ri[1];
    const double a3 = ri[2] * rj[1] - rj[2] * ri[1];
    const double a4 = ri[0] * rj[2] - rj[0] * ri[2];
    dadi_rotate(a1, a2, a3, a4, d);
}

// after the i-solve: T_zeta^-1 T_xi (residuals.F90:1540-1583)
__device__ __forceinline__ void dadi_post_i(const BlkView& b, long c, double d[5])
{
    double ri[3], rk[3];
    unit_mean_normal(b.sI, c, 1, b.nbox, ri);
    unit_mean_normal(b.sK, c, b.ldk, b.nbox, rk);
    const double a1 = ri[0] * rk[0] + ri[1] * rk[1] + ri[2] * rk[2];
    const double a2 = rk[0] * ri[1] - ri[0] * rk[1];
    const double a3 = rk[2] * ri[1] - ri[2] * rk[1];
    const double a4 = rk[0] * ri[2] - ri[0] * rk[2];
    dadi_rotate(a1, a2, a3, a4, d);
}

// after the k-solve: T_zeta and the -1/vol scaling (residuals.F90:1676-1745)
__device__ __forceinline__ void dadi_post_k(const BlkView& b, long c, double d[5], double gam)
{
    const long nb = b.nbox;
    const double rho = b.w[c], uvel = b.w[c + nb], vvel = b.w[c + 2 * nb], wvel = b.w[c + 3 * nb];
    const long s = b.ldk;
    double rk1 = 0.5 * (b.sK[c] + b.sK[c - s]), rk2 = 0.5 * (b.sK[c + nb] + b.sK[c - s + nb]),
           rk3 = 0.5 * (b.sK[c + 2 * nb] + b.sK[c - s + 2 * nb]);
    const double rk = sqrt(rk1 * rk1 + rk2 * rk2 + rk3 * rk3);
    const double uu = uvel * rk1 + vvel * rk2 + wvel * rk3;
    rk1 /= rk; rk2 /= rk; rk3 /= rk;
    const double uvw = 0.5 * (uvel * uvel + vvel * vvel + wvel * wvel);
    const double cijkinv = sqrt(rho / gam / b.p[c]);
    const double alph = rho * cijkinv * (1.0 / sqrt(2.0));
    const double xfact = 2.0 / cijkinv;
    const double ge = gam * b.w[c + 4 * nb] / rho - (gam - 1.0) * uvw;
    const double dw1 = d[0], dw2 = d[1], dw3 = d[2];
    const double dw4 = d[3] * alph, dw5 = d[4] * alph;
    const double a1 = dw1 * rk1 + dw2 * rk2 + dw3 * rk3 + dw4 + dw5;
    const double a2 = 0.5 * xfact * (dw4 - dw5);
    const double a3 = uvw * (rk1 * dw1 + rk2 * dw2 + rk3 * dw3);
    const double volfact = -1.0 / b.vol[c];
    d[0] = a1 * volfact;
    d[1] = (a1 * uvel - rho * (rk3 * dw2 - rk2 * dw3) + a2 * rk1) * volfact;
    d[2] = (a1 * vvel - rho * (rk1 * dw3 - rk3 * dw1) + a2 * rk2) * volfact;
    d[3] = (a1 * wvel - rho * (rk2 * dw1 - rk1 * dw2) + a2 * rk3) * volfact;
    d[4] = (a3 + rho * ((vvel * rk3 - wvel * rk2) * dw1 + (wvel * rk1 - uvel * rk3) * dw2 + (uvel * rk2 - vvel * rk1) * dw3) +
            (ge + 0.5 * xfact * uu / rk) * dw4 + (ge - 0.5 * xfact * uu / rk) * dw5) * volfact;
}

// One direction of the D-ADI sweep, one line per lane.
//  DIR 1 (j): pre-transform T_eta^-1, solve along j, post-transform T_xi^-1 T_eta
//  DIR 2 (k): solve along k, post T_zeta * (-1/vol)
// scale: factor applied to the incoming dw on load (DIR 1 only: -cfl*dtl*vol of
// executeDADIStep, smoothers.F90:514-532)
// POSTI (DIR 2, tiled i sweep in front): the transform T_zeta^-1 T_xi that follows the i-solve is applied to the update as it is
// loaded here instead of in a pointwise pass of its own (one read and one write of dw less)
// Round 4: both marches are software pipelines.  A lane walks its line alone (82 k lines per level: 1.5 wavefronts per SIMD), so
// nothing hides a load but the thread itself: the values of cell m+2 and the update of cell m+1 are REQUESTED in step m (two value
// sets alternate, the loop runs two steps per trip so that no register copy waits for a load), the coefficients of cell m+1 are formed
// from the set requested one step earlier, and the pre-transform of the j sweep reuses the values the coefficients were formed
// from; the back substitution requests row m-1 (update, eliminated super-diagonals, the metrics of its transform) in step m.
struct DadiPre { double rho, u, v, w, p, n0[3], nm[3], sc0; };      // what T_eta^-1 and the scaling of cell m need (from DadiRaw)
struct DadiBack { double d[5], sc[3], g[13]; };                      // row of the back substitution + the metrics of its transform

__device__ __forceinline__ void dadi_pre_of(const KParams& kp, const DadiRaw& r, DadiPre& o)
{
    o.rho = r.rho; o.u = r.u; o.v = r.v; o.w = r.w; o.p = r.p;
#pragma unroll
    for (int q = 0; q < 3; ++q) { o.n0[q] = r.n0[q]; o.nm[q] = r.nm[q]; }
    o.sc0 = -kp.cfl * r.dtl * r.vol;      // executeDADIStep scaling
}

// T_eta^-1 (dadi_pre_j) from values
__device__ __forceinline__ void dadi_pre_j_v(const DadiPre& q, double d[5], double gam)
{
    const double rho = q.rho, uvel = q.u, vvel = q.v, wvel = q.w;
    const double gm1 = gam - 1.0;
    const double orho = rcp_nr(rho);
    const double c2 = gam * q.p * orho;
    const double ocijk = rsq_nr(c2), cijk = c2 * ocijk;
    const double c2inv = ocijk * ocijk;
    const double oxfact = 0.5 * ocijk;                    // 1 / (2 c)
    const double alphinv = sqrt(2.0) * cijk * orho;
    const double uvw = 0.5 * (uvel * uvel + vvel * vvel + wvel * wvel);
    double rj1 = 0.5 * (q.n0[0] + q.nm[0]), rj2 = 0.5 * (q.n0[1] + q.nm[1]), rj3 = 0.5 * (q.n0[2] + q.nm[2]);
    const double orj = rsq_nr(rj1 * rj1 + rj2 * rj2 + rj3 * rj3);
    const double uu = uvel * rj1 + vvel * rj2 + wvel * rj3;
    rj1 *= orj; rj2 *= orj; rj3 *= orj;
    const double dw1 = d[0], dw2 = d[1], dw3 = d[2], dw4 = d[3], dw5 = d[4];
    double a1 = dw2 * uvel + dw3 * vvel + dw4 * wvel - dw5;
    a1 = a1 * gm1 * c2inv + dw1 * (1.0 - uvw * gm1 * c2inv);
    const double a2 = (rj2 * wvel - rj3 * vvel) * dw1 + rj3 * dw3 - rj2 * dw4;
    const double a3 = (rj3 * uvel - rj1 * wvel) * dw1 + rj1 * dw4 - rj3 * dw2;
    const double a4 = (rj1 * vvel - rj2 * uvel) * dw1 + rj2 * dw2 - rj1 * dw3;
    double a5 = uvw * dw1 - uvel * dw2 - vvel * dw3 - wvel * dw4 + dw5;
    a5 = a5 * gm1 * c2inv;
    const double a6 = uu * dw1 * orj - rj1 * dw2 - rj2 * dw3 - rj3 * dw4;
    d[0] = a1 * rj1 + a2 * orho;
    d[1] = a1 * rj2 + a3 * orho;
    d[2] = a1 * rj3 + a4 * orho;
    d[3] = (0.5 * a5 - a6 * oxfact) * alphinv;
    d[4] = (0.5 * a5 + a6 * oxfact) * alphinv;
}

// row m of the back substitution and what its transform reads: DIR 1: sI(c), sI(c-1), sJ(c), sJ(c-ldi); DIR 2: rho, u, v, w, p,
// rhoE, vol, sK(c), sK(c-ldk)
template <int DIR>
__device__ __forceinline__ void dadi_back_load(const BlkView& b, long c, bool withSc, const double* __restrict__ sc, DadiBack& o)
{
    static const int grp[5] = {0, 0, 0, 1, 2};
    (void)grp;
    const long nb = b.nbox;
#pragma unroll
    for (int l = 0; l < 5; ++l) o.d[l] = b.dw[c + l * nb];
#pragma unroll
    for (int g = 0; g < 3; ++g) o.sc[g] = withSc ? sc[c + g * nb] : 0.0;
    if (DIR == 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            o.g[q] = b.sI[c + q * nb]; o.g[3 + q] = b.sI[c - 1 + q * nb];
            o.g[6 + q] = b.sJ[c + q * nb]; o.g[9 + q] = b.sJ[c - b.ldi + q * nb];
        }
        o.g[12] = 0.0;
    } else {
        o.g[0] = b.w[c]; o.g[1] = b.w[c + nb]; o.g[2] = b.w[c + 2 * nb]; o.g[3] = b.w[c + 3 * nb];
        o.g[4] = b.p[c]; o.g[5] = b.w[c + 4 * nb]; o.g[6] = b.vol[c];
#pragma unroll
        for (int q = 0; q < 3; ++q) { o.g[7 + q] = b.sK[c + q * nb]; o.g[10 + q] = b.sK[c - b.ldk + q * nb]; }
    }
}

__device__ __forceinline__ void unit_mean_normal_v(const double* n0, const double* nm, double r[3])
{
    r[0] = 0.5 * (n0[0] + nm[0]);
    r[1] = 0.5 * (n0[1] + nm[1]);
    r[2] = 0.5 * (n0[2] + nm[2]);
    const double orr = rsq_nr(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    r[0] *= orr; r[1] *= orr; r[2] *= orr;
}

// the post-transforms of dadi_post_j / dadi_post_k from the values of dadi_back_load
template <int DIR>
__device__ __forceinline__ void dadi_back_post(const DadiBack& q, double d[5], double gam)
{
    if (DIR == 1) {
        double ri[3], rj[3];
        unit_mean_normal_v(q.g, q.g + 3, ri);
        unit_mean_normal_v(q.g + 6, q.g + 9, rj);
        const double a1 = ri[0] * rj[0] + ri[1] * rj[1] + ri[2] * rj[2];
        const double a2 = ri[0] * rj[1] - rj[0] * ri[1];
        const double a3 = ri[2] * rj[1] - rj[2] * ri[1];
        const double a4 = ri[0] * rj[2] - rj[0] * ri[2];
        dadi_rotate(a1, a2, a3, a4, d);
        return;
    }
    const double rho = q.g[0], uvel = q.g[1], vvel = q.g[2], wvel = q.g[3];
    double rk1 = 0.5 * (q.g[7] + q.g[10]), rk2 = 0.5 * (q.g[8] + q.g[11]), rk3 = 0.5 * (q.g[9] + q.g[12]);
    const double ork = rsq_nr(rk1 * rk1 + rk2 * rk2 + rk3 * rk3);
    const double uu = uvel * rk1 + vvel * rk2 + wvel * rk3;
    rk1 *= ork; rk2 *= ork; rk3 *= ork;
    const double uvw = 0.5 * (uvel * uvel + vvel * vvel + wvel * wvel);
    const double orho = rcp_nr(rho);
    const double c2 = gam * q.g[4] * orho;                 // c^2
    const double cijkinv = rsq_nr(c2);
    const double alph = rho * cijkinv * (1.0 / sqrt(2.0));
    const double xfact = 2.0 * c2 * cijkinv;               // 2 c
    const double ge = gam * q.g[5] * orho - (gam - 1.0) * uvw;
    const double dw1 = d[0], dw2 = d[1], dw3 = d[2];
    const double dw4 = d[3] * alph, dw5 = d[4] * alph;
    const double a1 = dw1 * rk1 + dw2 * rk2 + dw3 * rk3 + dw4 + dw5;
    const double a2 = 0.5 * xfact * (dw4 - dw5);
    const double a3 = uvw * (rk1 * dw1 + rk2 * dw2 + rk3 * dw3);
    const double volfact = -rcp_nr(q.g[6]);
    const double hx = 0.5 * xfact * uu * ork;
    d[0] = a1 * volfact;
    d[1] = (a1 * uvel - rho * (rk3 * dw2 - rk2 * dw3) + a2 * rk1) * volfact;
    d[2] = (a1 * vvel - rho * (rk1 * dw3 - rk3 * dw1) + a2 * rk2) * volfact;
    d[3] = (a1 * wvel - rho * (rk2 * dw1 - rk1 * dw2) + a2 * rk3) * volfact;
    d[4] = (a3 + rho * ((vvel * rk3 - wvel * rk2) * dw1 + (wvel * rk1 - uvel * rk3) * dw2 + (uvel * rk2 - vvel * rk1) * dw3) +
            (ge + hx) * dw4 + (ge - hx) * dw5) * volfact;
}

// UPD (DIR 2): the state update of executeDADIStep follows the last transform in the same step (no residual averaging in between):
// the back substitution holds the state of the cell already (it feeds T_zeta), so k_stage_update's read of dw, w, p is saved
// PIPE = false: the values of cell m+1 and the update of cell m are requested in step m itself (fewer registers: the wavefronts of a
// level then all fit the chip at once) -- measured against the pipeline per direction, see launch_dadi_level
template <int DIR, bool POSTI = false, bool UPD = false, bool PIPE = true>
__global__ __launch_bounds__(64) void k_dadi_sweep(const BlkView* __restrict__ tab, KParams kp, int slot0)
{
    static_assert(DIR == 1 || DIR == 2, "the i direction has its own kernels");
    static_assert(!UPD || DIR == 2, "the update follows the k sweep");
    const BlkView& b = tab[slot0 + blockIdx.z + 1];     // level-batched: one z-slice of the grid per block
    const int a = blockIdx.x * 64 + threadIdx.x + 2;
    const int bb = blockIdx.y + 2;
    int n, amax, bmax;
    long c0, s;
    const double* sN;
    if (DIR == 1) { amax = b.il; bmax = b.kl; n = b.ny; c0 = b.idx(a, 2, bb); s = b.ldi; sN = b.sJ; }
    else { amax = b.il; bmax = b.jl; n = b.nz; c0 = b.idx(a, bb, 2); s = b.ldk; sN = b.sK; }
    if (b.nx == 0 || a > amax || bb > bmax) return;
    const long nb = b.nbox;
    double* sc = b.scratch;   // components 0..2: modified super-diagonals of the three eigenvalue groups
    static const int grp[5] = {0, 0, 0, 1, 2};
    const double gam = kp.gammaConstant;
    if (n <= 1) {
        // "if (jl > 2)" etc.: no inversion for one-cell lines, the transforms only
        double d[5];
#pragma unroll
        for (int l = 0; l < 5; ++l) d[l] = b.dw[c0 + l * nb];
        if (DIR == 1) {
            const double sc0 = -kp.cfl * b.dtl[c0] * b.vol[c0];
#pragma unroll
            for (int l = 0; l < 5; ++l) d[l] *= sc0;
            dadi_pre_j(b, c0, d, gam);
            dadi_post_j(b, c0, d);
        } else {
            if (POSTI) dadi_post_i(b, c0, d);
            const double rho0 = b.w[c0], u0 = b.w[c0 + nb], v0 = b.w[c0 + 2 * nb], w0 = b.w[c0 + 3 * nb], e0 = b.w[c0 + 4 * nb], p0 = b.p[c0];
            dadi_post_k(b, c0, d, gam);
            if (UPD) stage_update_cell(b, kp, c0, d, rho0, u0, v0, w0, e0, p0, rho0, u0, v0, w0, p0);
        }
#pragma unroll
        for (int l = 0; l < 5; ++l) b.dw[c0 + l * nb] = d[l];
        return;
    }
    // ---- forward elimination.  State entering step m: cur = coefficients of cell m, prv of m-1, pre = transform values of cell m,
    // dX = update of cell m, rX = raw values of cell m+1 (requested a step ago); the step requests cell m+2 into (rY, dY)
    DadiCell cur, nxt, prv;
    DadiPre pre;
    DadiRaw rA, rB;
    double dA[5], dB[5];
    double ddp[3] = {0, 0, 0};          // dd'(m-1)
    double fprev[5] = {0, 0, 0, 0, 0};  // ff'(m-1)
    dadi_load<DIR>(b, kp, c0, s, sN, rA);
    dadi_coef<DIR>(kp, rA, cur);
    dadi_pre_of(kp, rA, pre);
    prv = cur;
#pragma unroll
    for (int l = 0; l < 5; ++l) dA[l] = b.dw[c0 + l * nb];
    dadi_load<DIR>(b, kp, c0 + s, s, sN, rA);          // cell 1 (n >= 2)
    auto fwd = [&](int m, double* __restrict__ dX, DadiRaw& rX, double* __restrict__ dY, DadiRaw& rY) {
        const long c = c0 + m * s;
        if (PIPE) {
            if (m + 2 < n) dadi_load<DIR>(b, kp, c + 2 * s, s, sN, rY);
            if (m + 1 < n) {
#pragma unroll
                for (int l = 0; l < 5; ++l) dY[l] = b.dw[c + s + l * nb];
            }
        } else if (m > 0 && m + 1 < n) {
            dadi_load<DIR>(b, kp, c + s, s, sN, rX);          // (cell 1 was requested in front of the loop)
        }
        double d[5];
#pragma unroll
        for (int l = 0; l < 5; ++l) d[l] = (PIPE || m == 0) ? dX[l] : b.dw[c + l * nb];
        if (DIR == 1) {
#pragma unroll
            for (int l = 0; l < 5; ++l) d[l] *= pre.sc0;
            dadi_pre_j_v(pre, d, gam);
        }
        if (DIR == 2 && POSTI) dadi_post_i(b, c, d);
        if (m < n - 1) { dadi_coef<DIR>(kp, rX, nxt); dadi_pre_of(kp, rX, pre); }
        double ddn[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            // row m: sub-diagonal from cell m-1, super-diagonal from cell m+1, both scaled with ddt(m)
            const double bbv = (m > 0) ? (-prv.vt1 - prv.dP[g]) * cur.ddt : 0.0;
            const double ddv = (m < n - 1) ? (-nxt.vt3 + nxt.dM[g]) * cur.ddt : 0.0;
            const double ccv = 1.0 + (cur.vt1 + cur.vt3 + cur.dP[g] - cur.dM[g]) * cur.ddt;
            const double d0 = rcp_nr(ccv - bbv * ddp[g]);
            ddn[g] = ddv * d0;
            sc[c + g * nb] = ddn[g];
#pragma unroll
            for (int l = 0; l < 5; ++l)
                if (grp[l] == g) d[l] = (d[l] - bbv * fprev[l]) * d0;
        }
#pragma unroll
        for (int g = 0; g < 3; ++g) ddp[g] = ddn[g];
#pragma unroll
        for (int l = 0; l < 5; ++l) fprev[l] = d[l];
        prv = cur;
        cur = nxt;
#pragma unroll
        for (int l = 0; l < 5; ++l) b.dw[c + l * nb] = d[l];
    };
    if (PIPE) {
        for (int m = 0; m < n; m += 2) {
            fwd(m, dA, rA, dB, rB);
            if (m + 1 < n) fwd(m + 1, dB, rB, dA, rA);
        }
    } else {
        for (int m = 0; m < n; ++m) fwd(m, dA, rA, dA, rA);
    }
    // ---- back substitution + post-transform; row n-1 keeps its value (fprev), its transform is applied like the others
    DadiBack qA, qB;
    dadi_back_load<DIR>(b, c0 + (long)(n - 1) * s, false, sc, qA);
    auto bwd = [&](int m, DadiBack& qX, DadiBack& qY) {
        const long c = c0 + m * s;
        if (m > 0) dadi_back_load<DIR>(b, c - s, true, sc, qY);
        double d[5];
        if (m < n - 1) {
#pragma unroll
            for (int l = 0; l < 5; ++l) d[l] = qX.d[l] - qX.sc[grp[l]] * fprev[l];
        } else {
#pragma unroll
            for (int l = 0; l < 5; ++l) d[l] = fprev[l];
        }
#pragma unroll
        for (int l = 0; l < 5; ++l) fprev[l] = d[l];
        dadi_back_post<DIR>(qX, d, gam);
#pragma unroll
        for (int l = 0; l < 5; ++l) b.dw[c + l * nb] = d[l];
        if (UPD) stage_update_cell(b, kp, c, d, qX.g[0], qX.g[1], qX.g[2], qX.g[3], qX.g[5], qX.g[4], qX.g[0], qX.g[1], qX.g[2], qX.g[3], qX.g[4]);
    };
    for (int m = n - 1; m >= 0; m -= 2) {
        bwd(m, qA, qB);
        if (m - 1 >= 0) bwd(m - 1, qB, qA);
    }
}

// ---------------------------------------------------------------------------
// i-direction D-ADI sweep in three coalesced pieces (same arithmetic as k_dadi_sweep<0>):
//   k_dadi_rows_i   pointwise, lanes over i: the tridiagonal rows (bb, cc, dd of the three
//                   eigenvalue groups) of every owned cell -> grad(0:8) (free between residuals:
//                   k_nodal_gradients rewrites it before every use)
//   k_dadi_solve_i  one thread per (line, equation): Thomas along i on 64-line x 16-cell LDS tiles,
//                   global traffic in 128-byte row segments (as k_res_averaging_i)
//   k_dadi_post_i   pointwise: T_zeta^-1 T_xi
// The one-line-per-lane form reads ~25 arrays with a stride of one row per lane: 0.9 ms per
// 128x128x96 block against 0.15 ms for the j sweep.
// ---------------------------------------------------------------------------
#define TI_SH 3                 // log2 of the chunk length: 8 cells -> 18 KB of LDS per workgroup, 8 workgroups per CU
#define TI_CH (1 << TI_SH)
#define TI_LD (TI_CH + 1)
#define TI_LPA (64 >> TI_SH)    // lines moved per wave access

// 64 lines (j0..j0+63) x TI_CH cells (i0..) of one array <-> LDS tile; lane -> (line TI_LPA*q + lane/TI_CH, cell lane%TI_CH)
__device__ __forceinline__ void tile_load(const BlkView& b, const double* __restrict__ arr, double* __restrict__ tile, int j0, int k,
                                          int i0, int lane)
{
    const int sub = lane >> TI_SH, col = lane & (TI_CH - 1), i = i0 + col;
#pragma unroll
    for (int q = 0; q < TI_CH; ++q) {
        const int r = TI_LPA * q + sub;
        tile[r * TI_LD + col] = (j0 + r <= b.jl && i <= b.il) ? arr[b.idx(i, j0 + r, k)] : 0.0;
    }
}

__device__ __forceinline__ void tile_store(const BlkView& b, double* __restrict__ arr, const double* __restrict__ tile, int j0, int k,
                                           int i0, int lane)
{
    const int sub = lane >> TI_SH, col = lane & (TI_CH - 1), i = i0 + col;
#pragma unroll
    for (int q = 0; q < TI_CH; ++q) {
        const int r = TI_LPA * q + sub;
        if (j0 + r <= b.jl && i <= b.il) arr[b.idx(i, j0 + r, k)] = tile[r * TI_LD + col];
    }
}

// A wavefront covers 64 consecutive cells of an i line of which the inner 62 produce rows: the coefficients of a cell (five square
// roots and eight divisions) are formed ONCE and reach the neighbouring rows by lane shifts (three evaluations per cell before: the
// kernel was bound by FP64 issue, 0.55 ms on 8 x 128x128x96)
#define DR_OUT 62
__global__ __launch_bounds__(SM_BX* SM_BY) void k_dadi_rows_i(const BlkView* __restrict__ tab, int nzb, KParams kp)
{
    const BlkView& b = tab[blockIdx.z / nzb + 1];
    const int lane = threadIdx.x;
    const int i = blockIdx.x * DR_OUT + lane + 1;              // lanes 1 .. 62 produce the cells 2 + 62 bx .. 63 + 62 bx
    const int j = blockIdx.y * SM_BY + threadIdx.y + 2;
    const int k = blockIdx.z % nzb + 2;
    if (b.nx == 0 || k > b.kl || (int)(blockIdx.x * DR_OUT) + 2 > b.il || (int)(blockIdx.y * SM_BY) + 2 > b.jl) return;   // uniform per workgroup
    const int ic = (i < b.ie) ? i : b.ie;                      // (cells 1 and ie: the neighbours of the first / last row, never used)
    const int jc = (j < b.jl) ? j : b.jl;                      // rows beyond the block repeat the last one and store nothing
    const long c = b.idx(ic, jc, k), nb = b.nbox;
    const int m = i - 2, n = b.nx;
    DadiCell cur;
    dadi_cell<0>(b, kp, c, 1, b.sI, cur);
    const double pvt1 = lane_up1(cur.vt1), nvt3 = lane_dn1(cur.vt3);
    double pdP[3], ndM[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) { pdP[g] = lane_up1(cur.dP[g]); ndM[g] = lane_dn1(cur.dM[g]); }
    if (lane < 1 || lane > DR_OUT || i > b.il || j > b.jl) return;
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const double bbv = (m > 0) ? (-pvt1 - pdP[g]) * cur.ddt : 0.0;
        const double ddv = (m < n - 1) ? (-nvt3 + ndM[g]) * cur.ddt : 0.0;
        const double ccv = 1.0 + (cur.vt1 + cur.vt3 + cur.dP[g] - cur.dM[g]) * cur.ddt;
        b.grad[c + (3 * g) * nb] = bbv;
        b.grad[c + (3 * g + 1) * nb] = ccv;
        b.grad[c + (3 * g + 2) * nb] = ddv;
    }
}

// Register form of the tile transposition: the global loads of a 64-line x TI_CH-cell tile land in registers
// (coalesced: lane -> (line TI_LPA*q + lane/TI_CH, cell lane%TI_CH)), pass through ONE LDS tile and come back as the
// TI_CH cells of the lane's own line.  All arrays of a chunk are fetched before the first transposition, so their
// latencies overlap, and a workgroup (one wave) needs 4.6 KB of LDS instead of one tile per array (18 KB: 8 waves per CU;
// the PMC counters showed 93 % of the wave cycles waiting).
__device__ __forceinline__ void tile_fetch(const BlkView& b, const double* __restrict__ arr, int j0, int k, int i0, int lane,
                                           double raw[TI_CH])
{
    const int sub = lane >> TI_SH, col = lane & (TI_CH - 1), i = i0 + col;
#pragma unroll
    for (int q = 0; q < TI_CH; ++q) {
        const int r = TI_LPA * q + sub;
        raw[q] = (j0 + r <= b.jl && i <= b.il) ? arr[b.idx(i, j0 + r, k)] : 0.0;
    }
}

// raw (coalesced order) -> v (cells 0..TI_CH-1 of the lane's line); one wave per workgroup: the barriers only order LDS
__device__ __forceinline__ void tile_to_line(double* __restrict__ tile, int lane, const double raw[TI_CH], double v[TI_CH])
{
    const int sub = lane >> TI_SH, col = lane & (TI_CH - 1);
#pragma unroll
    for (int q = 0; q < TI_CH; ++q) tile[(TI_LPA * q + sub) * TI_LD + col] = raw[q];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < TI_CH; ++m) v[m] = tile[lane * TI_LD + m];
    __syncthreads();
}

__device__ __forceinline__ void line_to_global(const BlkView& b, double* __restrict__ arr, double* __restrict__ tile, int j0, int k,
                                               int i0, int lane, const double v[TI_CH])
{
#pragma unroll
    for (int m = 0; m < TI_CH; ++m) tile[lane * TI_LD + m] = v[m];
    __syncthreads();
    tile_store(b, arr, tile, j0, k, i0, lane);
    __syncthreads();
}

__global__ __launch_bounds__(64) void k_dadi_solve_i(const BlkView* __restrict__ tab, KParams kp)
{
    __shared__ double tile[64 * TI_LD];
    const BlkView& b = tab[blockIdx.z + 1];
    const int l = blockIdx.x % 5;               // equation fastest: the three equations of group 0 share their rows in L2
    const int g = (l < 3) ? 0 : l - 2;
    const int lane = threadIdx.x;
    const int j0 = (blockIdx.x / 5) * 64 + 2, k = blockIdx.y + 2;
    const int n = b.nx;
    if (b.nx == 0 || j0 > b.jl || k > b.kl || n <= 1) return;
    const long nb = b.nbox;
    const double* __restrict__ Bb = b.grad + (3 * g) * nb;
    const double* __restrict__ Cc = b.grad + (3 * g + 1) * nb;
    const double* __restrict__ Dd = b.grad + (3 * g + 2) * nb;
    double* __restrict__ Dp = b.scratch + l * nb;       // eliminated super-diagonal of this equation's solve
    double* __restrict__ F = b.dw + l * nb;
    const int nch = (n + TI_CH - 1) / TI_CH;
    double ddp = 0.0, fprev = 0.0;
    for (int ch = 0; ch < nch; ++ch) {
        const int i0 = 2 + ch * TI_CH;
        double rb[TI_CH], rc[TI_CH], rd[TI_CH], rf[TI_CH];
        tile_fetch(b, Bb, j0, k, i0, lane, rb); tile_fetch(b, Cc, j0, k, i0, lane, rc);
        tile_fetch(b, Dd, j0, k, i0, lane, rd); tile_fetch(b, F, j0, k, i0, lane, rf);
        double vb[TI_CH], vc[TI_CH], vd[TI_CH], vf[TI_CH];
        tile_to_line(tile, lane, rb, vb); tile_to_line(tile, lane, rc, vc);
        tile_to_line(tile, lane, rd, vd); tile_to_line(tile, lane, rf, vf);
        const int mEnd = (n - ch * TI_CH < TI_CH) ? n - ch * TI_CH : TI_CH;
#pragma unroll
        for (int m = 0; m < TI_CH; ++m) {
            if (m < mEnd) {
                const double bbv = vb[m];
                const double d0 = 1.0 / (vc[m] - bbv * ddp);
                const double ddn = vd[m] * d0;
                vd[m] = ddn;
                const double f = (vf[m] - bbv * fprev) * d0;
                vf[m] = f;
                ddp = ddn; fprev = f;
            }
        }
        line_to_global(b, Dp, tile, j0, k, i0, lane, vd);
        line_to_global(b, F, tile, j0, k, i0, lane, vf);
    }
    // back substitution: rows n-2 .. 0 (row n-1 keeps its value = fprev)
    for (int ch = nch - 1; ch >= 0; --ch) {
        const int i0 = 2 + ch * TI_CH;
        double rd[TI_CH], rf[TI_CH], vd[TI_CH], vf[TI_CH];
        tile_fetch(b, Dp, j0, k, i0, lane, rd); tile_fetch(b, F, j0, k, i0, lane, rf);
        tile_to_line(tile, lane, rd, vd); tile_to_line(tile, lane, rf, vf);
        int mTop = n - 2 - ch * TI_CH;
        if (mTop > TI_CH - 1) mTop = TI_CH - 1;
#pragma unroll
        for (int m = TI_CH - 1; m >= 0; --m) {
            if (m <= mTop) {
                const double f = vf[m] - vd[m] * fprev;
                vf[m] = f;
                fprev = f;
            }
        }
        line_to_global(b, F, tile, j0, k, i0, lane, vf);
    }
}

__global__ __launch_bounds__(SM_BX* SM_BY) void k_dadi_post_i(const BlkView* __restrict__ tab, int nzb)
{
    const BlkView& b = tab[blockIdx.z / nzb + 1];
    const int i = blockIdx.x * SM_BX + threadIdx.x + 2;
    const int j = blockIdx.y * SM_BY + threadIdx.y + 2;
    const int k = blockIdx.z % nzb + 2;
    if (i > b.il || j > b.jl || k > b.kl) return;
    const long c = b.idx(i, j, k), nb = b.nbox;
    double d[5];
#pragma unroll
    for (int l = 0; l < 5; ++l) d[l] = b.dw[c + l * nb];
    dadi_post_i(b, c, d);
#pragma unroll
    for (int l = 0; l < 5; ++l) b.dw[c + l * nb] = d[l];
}


// ---------------------------------------------------------------------------
// Round 4: the i direction of D-ADI as PARALLEL CYCLIC REDUCTION along the lanes.  The three pieces above move every value four
// times through transposing tiles (counted: rows 221 + solve 557 B per cell and sub-iteration, profiles/r04_e_config3_pmc_bytes.md).
// An i line lies along the lanes: a workgroup of NW wavefronts holds one whole line (nx <= 64 NW cells), every thread forms the
// tridiagonal rows of ITS cell from coalesced loads (dadi_cell<0>; the neighbours' terms through LDS), and the three coefficient sets
// with their five right-hand sides are reduced in ceil(log2 nx) steps, all lanes busy: step s eliminates the unknowns at distance s,
//   alpha = -a_i, gamma = -c_i (rows kept normalised: b = 1),  b' = 1 + alpha c_(i-s) + gamma a_(i+s),
//   a' = alpha a_(i-s) / b',  c' = gamma c_(i+s) / b',  d' = (d + alpha d_(i-s) + gamma d_(i+s)) / b'
// until every row stands alone (x = d).  The rows are diagonally dominant (cc = 1 + positive terms), so the reduction is as stable as
// the Thomas recurrence of the reference (residuals.F90:1750-1783); results agree to rounding.  The transform that follows the
// i-solve (T_zeta^-1 T_xi, residuals.F90:1540-1583) is applied before the store: the k sweep loads a finished update.
// One read of dw + the cell data, one write of dw; no scratch arrays.  A workgroup walks PI_JL consecutive j lines of one k plane.
// ---------------------------------------------------------------------------
#define PI_JL 8
#define PI_NC 11          // components in LDS per cell and buffer: a(3), c(3), d(5)

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_dadi_i_pcr(const BlkView* __restrict__ tab, KParams kp)
{
    constexpr int T = 64 * NW;
    __shared__ double P[PI_NC * T];      // ONE buffer, two barriers per step: half the LDS = twice the workgroups per CU (the kernel is
                                         // bound by latency: 3 -> 6 wavefronts per SIMD, 0.72 -> 0.66 ms)
    const BlkView& b = tab[blockIdx.z + 1];
    const int t = threadIdx.x, n = b.nx;
    const int k = blockIdx.y + 2;
    const int j0 = blockIdx.x * PI_JL + 2;
    if (b.nx == 0 || k > b.kl || j0 > b.jl || n > T) return;        // uniform per workgroup
    const bool act = t < n;
    const int tc = act ? t : n - 1;                                  // inactive threads repeat the last cell (loads stay inside the box)
    const long nb = b.nbox;
    static const int grp[5] = {0, 0, 0, 1, 2};
    double* __restrict__ X = P;
    for (int jl_ = 0; jl_ < PI_JL; ++jl_) {
        const int j = j0 + jl_;
        if (j > b.jl) break;                                         // uniform
        const long c = b.idx(2 + tc, j, k);
        double d[5];
#pragma unroll
        for (int l = 0; l < 5; ++l) d[l] = b.dw[c + l * nb];
        double a[3], cc[3];
        if (n > 1) {
            DadiCell cur;
            dadi_cell<0>(b, kp, c, 1, b.sI, cur);
            // (vt1, dP) go to the row above (cell i+1 reads its lower neighbour), (vt3, dM) to the row below
            X[0 * T + t] = cur.vt1; X[1 * T + t] = cur.dP[0]; X[2 * T + t] = cur.dP[1]; X[3 * T + t] = cur.dP[2];
            X[4 * T + t] = cur.vt3; X[5 * T + t] = cur.dM[0]; X[6 * T + t] = cur.dM[1]; X[7 * T + t] = cur.dM[2];
            __syncthreads();
            const int tm = (t > 0) ? t - 1 : 0, tp = (t < T - 1) ? t + 1 : T - 1;
            const double pvt1 = X[0 * T + tm], nvt3 = X[4 * T + tp];
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const double bbv = (t > 0) ? (-pvt1 - X[(1 + g) * T + tm]) * cur.ddt : 0.0;
                const double ddv = (t < n - 1) ? (-nvt3 + X[(5 + g) * T + tp]) * cur.ddt : 0.0;
                const double ccv = 1.0 + (cur.vt1 + cur.vt3 + cur.dP[g] - cur.dM[g]) * cur.ddt;
                const double inv = act ? rcp_nr(ccv) : 0.0;
                a[g] = bbv * inv; cc[g] = ddv * inv;                 // inactive threads: the identity row (a = c = 0, d = 0)
#pragma unroll
                for (int l = 0; l < 5; ++l)
                    if (grp[l] == g) d[l] = act ? d[l] * inv : 0.0;
            }
            __syncthreads();                                         // every thread has read its neighbours' exchange values
            for (int st = 1; st < n; st <<= 1) {
                double* __restrict__ Q = P;
#pragma unroll
                for (int g = 0; g < 3; ++g) { Q[g * T + t] = a[g]; Q[(3 + g) * T + t] = cc[g]; }
#pragma unroll
                for (int l = 0; l < 5; ++l) Q[(6 + l) * T + t] = d[l];
                __syncthreads();
                const bool lo = t >= st, hi = t + st < n;
                const int im = lo ? t - st : t, ip = hi ? t + st : t;
                double al[3], ga[3], inv[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const double am = lo ? Q[g * T + im] : 0.0, cm = lo ? Q[(3 + g) * T + im] : 0.0;
                    const double ap = hi ? Q[g * T + ip] : 0.0, cp = hi ? Q[(3 + g) * T + ip] : 0.0;
                    al[g] = -a[g]; ga[g] = -cc[g];
                    inv[g] = rcp_nr(1.0 + al[g] * cm + ga[g] * ap);
                    a[g] = al[g] * am * inv[g];
                    cc[g] = ga[g] * cp * inv[g];
                }
#pragma unroll
                for (int l = 0; l < 5; ++l) {
                    const int g = grp[l];
                    const double dm = lo ? Q[(6 + l) * T + im] : 0.0, dp = hi ? Q[(6 + l) * T + ip] : 0.0;
                    d[l] = (d[l] + al[g] * dm + ga[g] * dp) * inv[g];
                }
                __syncthreads();                                     // reads of this step done before the next step's writes
            }
        }
        if (act) {
            dadi_post_i(b, c, d);
#pragma unroll
            for (int l = 0; l < 5; ++l) b.dw[c + l * nb] = d[l];
        }
    }
}

template <int NW>
static void launch_dadi_i_pcr(const BlkView* tab, int nslots, int ny, int nz, const KParams& kp, hipStream_t s)
{
    hipLaunchKernelGGL((k_dadi_i_pcr<NW>), dim3((ny + PI_JL - 1) / PI_JL, nz, nslots), dim3(64 * NW, 1, 1), 0, s, tab, kp);
}

// (the k sweep runs as a software pipeline, the j sweep plain with the reciprocal forms: pipelined it needs 322 registers -- one wavefront
// per SIMD, its 1536 wavefronts in two rounds: 0.90 against 0.73 ms, round 4)
int g_dadi_pcr = 1;      // tuning "dadi_pcr": the i direction of D-ADI by cyclic reduction along the lanes (0: rows + tiled Thomas)

// computedwDADI incl. the -cfl*dtl*vol scaling of executeDADIStep
// withUpdate: the k sweep also updates the state (no residual averaging follows): finish_stage then skips k_stage_update
void launch_dadi_level(const BlkView* tab, int nslots, int nx, int ny, int nz, const KParams& kp, hipStream_t s, bool withUpdate)
{
    LEVEL_SPLIT(nslots, nz + 4, launch_dadi_level(tab + s0_, n_, nx, ny, nz, kp, s, withUpdate));
    if (nslots <= 0) return;
    dim3 blk(64, 1, 1);
    hipLaunchKernelGGL((k_dadi_sweep<1, false, false, false>), dim3((nx + 63) / 64, nz, nslots), blk, 0, s, tab, kp, 0);
    if (g_dadi_pcr && nx <= 256) {
        // i direction: cyclic reduction along the lanes with its transform applied (k_dadi_i_pcr); nx = the widest block of the level
        if (nx <= 64) launch_dadi_i_pcr<1>(tab, nslots, ny, nz, kp, s);
        else if (nx <= 128) launch_dadi_i_pcr<2>(tab, nslots, ny, nz, kp, s);
        else if (nx <= 192) launch_dadi_i_pcr<3>(tab, nslots, ny, nz, kp, s);
        else launch_dadi_i_pcr<4>(tab, nslots, ny, nz, kp, s);
        const dim3 gk((nx + 63) / 64, ny, nslots);
        if (withUpdate) hipLaunchKernelGGL((k_dadi_sweep<2, false, true>), gk, blk, 0, s, tab, kp, 0);
        else hipLaunchKernelGGL((k_dadi_sweep<2, false>), gk, blk, 0, s, tab, kp, 0);
        return;
    }
    // i direction: rows pointwise, Thomas per (line, equation) through LDS tiles; the transform behind the i-solve is applied
    // by the k sweep as it loads the update
    if (nx > 1) {
        const dim3 pg((nx + DR_OUT - 1) / DR_OUT, (ny + SM_BY - 1) / SM_BY, nz * nslots), pb(SM_BX, SM_BY, 1);
        hipLaunchKernelGGL(k_dadi_rows_i, pg, pb, 0, s, tab, nz, kp);
        hipLaunchKernelGGL(k_dadi_solve_i, dim3(5 * ((ny + 63) / 64), nz, nslots), blk, 0, s, tab, kp);
    }
    if (withUpdate) hipLaunchKernelGGL((k_dadi_sweep<2, true, true>), dim3((nx + 63) / 64, ny, nslots), blk, 0, s, tab, kp, 0);
    else hipLaunchKernelGGL((k_dadi_sweep<2, true>), dim3((nx + 63) / 64, ny, nslots), blk, 0, s, tab, kp, 0);
}
