// Multigrid transfer operators.
//
// Reference semantics:
//   transferToCoarseGrid   src/solver/multiGrid.F90:5-324   (restriction + forcing term)
//   transferToFineGrid     src/solver/multiGrid.F90:326-652 (trilinear prolongation of corrections)
// Both are gathers: a coarse cell reads its 8 fine cells (mgIFine/JFine/KFine), a
// fine cell reads its 8 nearest coarse cells (mgICoarse/...).  The reference
// aliases the coarse-level work arrays onto the level-1 block (SURVEY.md §7
// quirk 12); here every level owns private arrays, which consumes no stale value.
// Roofline: HBM; no MFMA.
#include "internal.h"

#define MG_BX 64
#define MG_BY 4

// coarse owned cells: wr = weighted sum of the 8 fine residuals; w, p, rev =
// volume-weighted averages; then Etot and the laminar viscosity of the coarse
// state (multiGrid.F90:92-226).  rev keeps the restricted value on coarse levels.
__global__ __launch_bounds__(MG_BX* MG_BY) void k_restrict(const BlkView* __restrict__ ctab, const BlkView* __restrict__ ftab, int nzb, KParams kp)
{
    const BlkView& c = ctab[blockIdx.z / nzb + 1];
    const BlkView& f = ftab[blockIdx.z / nzb + 1];
    const int i = blockIdx.x * MG_BX + threadIdx.x + 2;
    const int j = blockIdx.y * MG_BY + threadIdx.y + 2;
    const int k = blockIdx.z % nzb + 2;
    if (i > c.il || j > c.jl || k > c.kl) return;
    const long cc = c.idx(i, j, k);
    const int ii = c.mgIFine[2 * i], ii1 = c.mgIFine[2 * i + 1];
    const int jj = c.mgJFine[2 * j], jj1 = c.mgJFine[2 * j + 1];
    const int kk = c.mgKFine[2 * k], kk1 = c.mgKFine[2 * k + 1];
    const double weight = c.mgKWeight[k] * c.mgJWeight[j] * c.mgIWeight[i];
    // reference order of the eight fine cells in the state averages:
    // (ii,jj,kk) (ii,jj1,kk) (ii1,jj,kk) (ii1,jj1,kk) (ii,jj,kk1) (ii,jj1,kk1) (ii1,jj,kk1) (ii1,jj1,kk1)
    const long q[8] = {f.idx(ii, jj, kk),  f.idx(ii, jj1, kk),  f.idx(ii1, jj, kk),  f.idx(ii1, jj1, kk),
                       f.idx(ii, jj, kk1), f.idx(ii, jj1, kk1), f.idx(ii1, jj, kk1), f.idx(ii1, jj1, kk1)};
    double v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = f.vol[q[m]];
    // vola: (ii,jj,kk)+(ii1,jj,kk)+(ii,jj1,kk)+(ii1,jj1,kk)+(ii,jj,kk1)+(ii1,jj,kk1)+(ii,jj1,kk1)+(ii1,jj1,kk1)
    double vola = v[0] + v[2] + v[1] + v[3] + v[4] + v[6] + v[5] + v[7];
    vola = 1.0 / vola;
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        const double* d = f.dw + l * f.nbox;
        c.wr[cc + l * c.nbox] = (d[q[0]] + d[q[1]] + d[q[2]] + d[q[3]] + d[q[4]] + d[q[5]] + d[q[6]] + d[q[7]]) * weight;
    }
    double wv[4];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const double* a = f.w + l * f.nbox;
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < 8; ++m) s += v[m] * a[q[m]];
        wv[l] = s * vola;
        c.w[cc + l * c.nbox] = wv[l];
    }
    double sp = 0.0, sr = 0.0;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        sp += v[m] * f.p[q[m]];
        sr += v[m] * f.rev[q[m]];
    }
    const double pc = sp * vola;
    c.p[cc] = pc;
    c.rev[cc] = sr * vola;
    // computeEtotBlock + computeLamViscosity(.False.) on the coarse state
    const double ovgm1 = 1.0 / (kp.gammaConstant - 1.0);
    c.w[cc + 4 * c.nbox] = ovgm1 * pc + 0.5 * wv[0] * (wv[1] * wv[1] + wv[2] * wv[2] + wv[3] * wv[3]);
    if (kp.viscous) {
        const double muSuth = kp.muSuthDim / kp.muRef, TSuth = kp.TSuthDim / kp.TRef, SSuth = kp.SSuthDim / kp.TRef;
        const double T = pc / (kp.RGas * wv[0]);
        const double tt = T / TSuth;
        c.rlv[cc] = muSuth * ((TSuth + SSuth) / (T + SSuth)) * (tt * sqrt(tt));
    }
}

// cells 1..ie: w1 = w, p1 = p  (multiGrid.F90:258-276)
__global__ __launch_bounds__(MG_BX* MG_BY) void k_store_entry_state(const BlkView* __restrict__ ctab, int nzb)
{
    const BlkView& c = ctab[blockIdx.z / nzb + 1];
    const int i = blockIdx.x * MG_BX + threadIdx.x + (2 - 16);
    const int j = blockIdx.y * MG_BY + threadIdx.y + 1;
    const int k = blockIdx.z % nzb + 1;
    if (c.nx == 0 || i < 1 || i > c.ie || j > c.je || k > c.ke) return;
    const long cc = c.idx(i, j, k);
#pragma unroll
    for (int l = 0; l < 5; ++l) c.w1[cc + l * c.nbox] = c.w[cc + l * c.nbox];
    c.p1[cc] = c.p[cc];
}

// forcing term: tmp = fcoll*wr ; wr = tmp - dw ; dw = tmp  (multiGrid.F90:302-320)
__global__ __launch_bounds__(MG_BX* MG_BY) void k_forcing(const BlkView* __restrict__ ctab, int nzb, double fcoll)
{
    const BlkView& c = ctab[blockIdx.z / nzb + 1];
    const int i = blockIdx.x * MG_BX + threadIdx.x + 2;
    const int j = blockIdx.y * MG_BY + threadIdx.y + 2;
    const int k = blockIdx.z % nzb + 2;
    if (i > c.il || j > c.jl || k > c.kl) return;
    const long cc = c.idx(i, j, k);
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        const double tmp = fcoll * c.wr[cc + l * c.nbox];
        c.wr[cc + l * c.nbox] = tmp - c.dw[cc + l * c.nbox];
        c.dw[cc + l * c.nbox] = tmp;
    }
}

// corrections on the coarse block, cells 1..ie, into scratch(0:4):
// (rho,u,v,w) - w1 and p - p1  (multiGrid.F90:392-404; the reference overwrites w)
__global__ __launch_bounds__(MG_BX* MG_BY) void k_corrections(const BlkView* __restrict__ ctab, int nzb)
{
    const BlkView& c = ctab[blockIdx.z / nzb + 1];
    const int i = blockIdx.x * MG_BX + threadIdx.x + (2 - 16);
    const int j = blockIdx.y * MG_BY + threadIdx.y + 1;
    const int k = blockIdx.z % nzb + 1;
    if (c.nx == 0 || i < 1 || i > c.ie || j > c.je || k > c.ke) return;
    const long cc = c.idx(i, j, k);
#pragma unroll
    for (int l = 0; l < 4; ++l) c.scratch[cc + l * c.nbox] = c.w[cc + l * c.nbox] - c.w1[cc + l * c.nbox];
    c.scratch[cc + 4 * c.nbox] = c.p[cc] - c.p1[cc];
}

// fine owned cells: trilinear (27,9,3,1)/64 interpolation of the coarse corrections,
// state update with clipping, Etot, viscosities (multiGrid.F90:481-575)
__global__ __launch_bounds__(MG_BX* MG_BY) void k_prolong_update(const BlkView* __restrict__ ftab, const BlkView* __restrict__ ctab, int nzb, KParams kp)
{
    const BlkView& f = ftab[blockIdx.z / nzb + 1];
    const BlkView& c = ctab[blockIdx.z / nzb + 1];
    const int i = blockIdx.x * MG_BX + threadIdx.x + 2;
    const int j = blockIdx.y * MG_BY + threadIdx.y + 2;
    const int k = blockIdx.z % nzb + 2;
    if (i > f.il || j > f.jl || k > f.kl) return;
    const long cf = f.idx(i, j, k);
    const int ii = f.mgICoarse[2 * i], ii1 = f.mgICoarse[2 * i + 1];
    const int jj = f.mgJCoarse[2 * j], jj1 = f.mgJCoarse[2 * j + 1];
    const int kk = f.mgKCoarse[2 * k], kk1 = f.mgKCoarse[2 * k + 1];
    double d[5];
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        const double* ww = c.scratch + l * c.nbox;
        d[l] = 0.421875 * ww[c.idx(ii, jj, kk)] +
               0.140625 * (ww[c.idx(ii1, jj, kk)] + ww[c.idx(ii, jj1, kk)] + ww[c.idx(ii, jj, kk1)]) +
               0.046875 * (ww[c.idx(ii1, jj1, kk)] + ww[c.idx(ii1, jj, kk1)] + ww[c.idx(ii, jj1, kk1)]) +
               0.015625 * ww[c.idx(ii1, jj1, kk1)];
        f.dw[cf + l * f.nbox] = d[l];      // the reference leaves the corrections in dw
    }
    const long nb = f.nbox;
    double rho = f.w[cf] + d[0];
    const double u = f.w[cf + nb] + d[1], v = f.w[cf + 2 * nb] + d[2], w = f.w[cf + 3 * nb] + d[3];
    double p = f.p[cf] + d[4];
    rho = fmax(rho, 1.e-4 * kp.rhoInf);
    p = fmax(p, 1.e-4 * kp.pInfCorr);
    f.w[cf] = rho; f.w[cf + nb] = u; f.w[cf + 2 * nb] = v; f.w[cf + 3 * nb] = w;
    f.p[cf] = p;
    const double ovgm1 = 1.0 / (kp.gammaConstant - 1.0);
    f.w[cf + 4 * nb] = ovgm1 * p + 0.5 * rho * (u * u + v * v + w * w);
    if (kp.viscous) {
        const double muSuth = kp.muSuthDim / kp.muRef, TSuth = kp.TSuthDim / kp.TRef, SSuth = kp.SSuthDim / kp.TRef;
        const double T = p / (kp.RGas * rho);
        const double tt = T / TSuth;
        const double rlv = muSuth * ((TSuth + SSuth) / (T + SSuth)) * (tt * sqrt(tt));
        f.rlv[cf] = rlv;
        if (kp.eddyModel && kp.updateEddy) {
            const double cv13 = kp.sa_cv1 * kp.sa_cv1 * kp.sa_cv1;
            const double rnuSA = f.w[cf + 5 * nb] * rho;
            const double chi = rnuSA / rlv;
            const double chi3 = chi * chi * chi;
            f.rev[cf] = chi3 / (chi3 + cv13) * rnuSA;
        }
    }
}

// Level-batched launchers: one launch covers every block (pair) of the level, blockIdx.z = slot * nzb + plane;
// nx, ny, nz = the largest extents of the blocks the grid runs over (coarse blocks unless stated otherwise).
void launch_restrict_level(const BlkView* ctab, const BlkView* ftab, int nslots, int nx, int ny, int nz, const KParams& kp, hipStream_t s)
{
    LEVEL_SPLIT(nslots, nz + 4, launch_restrict_level(ctab + s0_, ftab + s0_, n_, nx, ny, nz, kp, s));
    if (nslots <= 0) return;
    hipLaunchKernelGGL(k_restrict, dim3((nx + MG_BX - 1) / MG_BX, (ny + MG_BY - 1) / MG_BY, nz * nslots), dim3(MG_BX, MG_BY, 1), 0, s,
                       ctab, ftab, nz, kp);
}

void launch_store_entry_state_level(const BlkView* ctab, int nslots, int nx, int ny, int nz, hipStream_t s)
{
    LEVEL_SPLIT(nslots, nz + 4, launch_store_entry_state_level(ctab + s0_, n_, nx, ny, nz, s));
    if (nslots <= 0) return;
    const int nzb = nz + 2;
    hipLaunchKernelGGL(k_store_entry_state, dim3((nx + 2 + 15 + MG_BX - 1) / MG_BX, (ny + 2 + MG_BY - 1) / MG_BY, nzb * nslots),
                       dim3(MG_BX, MG_BY, 1), 0, s, ctab, nzb);
}

void launch_forcing_level(const BlkView* ctab, int nslots, int nx, int ny, int nz, double fcoll, hipStream_t s)
{
    LEVEL_SPLIT(nslots, nz + 4, launch_forcing_level(ctab + s0_, n_, nx, ny, nz, fcoll, s));
    if (nslots <= 0) return;
    hipLaunchKernelGGL(k_forcing, dim3((nx + MG_BX - 1) / MG_BX, (ny + MG_BY - 1) / MG_BY, nz * nslots), dim3(MG_BX, MG_BY, 1), 0, s,
                       ctab, nz, fcoll);
}

void launch_corrections_level(const BlkView* ctab, int nslots, int nx, int ny, int nz, hipStream_t s)
{
    LEVEL_SPLIT(nslots, nz + 4, launch_corrections_level(ctab + s0_, n_, nx, ny, nz, s));
    if (nslots <= 0) return;
    const int nzb = nz + 2;
    hipLaunchKernelGGL(k_corrections, dim3((nx + 2 + 15 + MG_BX - 1) / MG_BX, (ny + 2 + MG_BY - 1) / MG_BY, nzb * nslots),
                       dim3(MG_BX, MG_BY, 1), 0, s, ctab, nzb);
}

// nx, ny, nz: largest FINE block
void launch_prolong_update_level(const BlkView* ftab, const BlkView* ctab, int nslots, int nx, int ny, int nz, const KParams& kp,
                                 hipStream_t s)
{
    LEVEL_SPLIT(nslots, nz + 4, launch_prolong_update_level(ftab + s0_, ctab + s0_, n_, nx, ny, nz, kp, s));
    if (nslots <= 0) return;
    hipLaunchKernelGGL(k_prolong_update, dim3((nx + MG_BX - 1) / MG_BX, (ny + MG_BY - 1) / MG_BY, nz * nslots), dim3(MG_BX, MG_BY, 1), 0,
                       s, ftab, ctab, nz, kp);
}
