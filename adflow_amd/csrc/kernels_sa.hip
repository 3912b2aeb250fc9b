// Spalart-Allmaras residual, fused per cell: source + upwind advection +
// diffusion + scaling -> dw(:,:,:,itu1).
//
// Reference semantics (block path, resOnly):
//   sa_block        src/turbulence/sa.F90:16-86
//   saSource        src/turbulence/sa.F90:89-344
//   turbAdvection   src/turbulence/turbUtils.F90:828-1561
//   saViscous       src/turbulence/sa.F90:346-676
//   saResScale      src/turbulence/sa.F90:678-715
// The reference accumulates in scratch(idvt) over four sweeps; here the value
// lives in a register.  Roofline: HBM; no MFMA.
#include "sa_core.h"

#define SA_BX 64
#define SA_BY 4

// SOLVE: additionally store the right-hand side (scratch 0) and the central
// jacobian qq (scratch 1) for the DDADI line solves of saSolve
template <bool SOLVE>
__global__ __launch_bounds__(SA_BX* SA_BY) void k_sa_residual(const BlkView* __restrict__ tab, int nzb, KParams kp)
{
    const BlkView& b = tab[blockIdx.z / nzb + 1];     // level-batched: blockIdx.z = slot * nzb + plane
    const int i = blockIdx.x * SA_BX + threadIdx.x + 2;
    const int j = blockIdx.y * SA_BY + threadIdx.y + 2;
    const int k = blockIdx.z % nzb + 2;
    if (i > b.il || j > b.jl || k > b.kl) return;
    const long c = b.idx(i, j, k);
    const long nb = b.nbox;
    const long si = 1, sj = b.ldi, sk = b.ldk;

    SaDir di, dj, dk;
    load_dir(b, c, si, b.sI, di, 0);
    load_dir(b, c, sj, b.sJ, dj, 1);
    load_dir(b, c, sk, b.sK, dk, 2);

    const double rho = b.w[c], u = b.w[c + nb], v = b.w[c + 2 * nb], w = b.w[c + 3 * nb];
    const double nut = dk.nt[2];
    const adf_real8 vol0 = b.vol[c];

    // ---- source (sa.F90:133-300): velocity gradient * 2 vol from the six neighbours
    double gu[3][3];   // gu[comp][xyz]
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const long off = (m + 1) * nb;
        const double qip = b.w[c + si + off], qim = b.w[c - si + off];
        const double qjp = b.w[c + sj + off], qjm = b.w[c - sj + off];
        const double qkp = b.w[c + sk + off], qkm = b.w[c - sk + off];
#pragma unroll
        for (int d = 0; d < 3; ++d)
            gu[m][d] = qip * di.sp[d] - qim * di.sm[d] + qjp * dj.sp[d] - qjm * dj.sm[d] + qkp * dk.sp[d] - qkm * dk.sm[d];
    }
    const adf_real8 fact = 0.25 / vol0;
    double ss, strainMag2 = 0.0;
    if (kp.turbProd == ADFLOW_TURBPROD_STRAIN) {
        const double sxx = 2.0 * fact * gu[0][0], syy = 2.0 * fact * gu[1][1], szz = 2.0 * fact * gu[2][2];
        const double sxy = fact * (gu[0][1] + gu[1][0]), sxz = fact * (gu[0][2] + gu[2][0]), syz = fact * (gu[1][2] + gu[2][1]);
        const double tr = sxx + syy + szz;
        const double div2 = (2.0 * (1.0 / 3.0)) * (tr * tr);
        strainMag2 = 2.0 * (sxy * sxy + sxz * sxz + syz * syz) + sxx * sxx + syy * syy + szz * szz;
        ss = sqrt(2.0 * strainMag2 - div2);
    } else {
        const double vortx = 2.0 * fact * (gu[2][1] - gu[1][2]);   // wheel speed omega = 0 (non-rotating sections)
        const double vorty = 2.0 * fact * (gu[0][2] - gu[2][0]);
        const double vortz = 2.0 * fact * (gu[1][0] - gu[0][1]);
        ss = sqrt(vortx * vortx + vorty * vorty + vortz * vortz);
    }
    const double cv13 = kp.sa_cv1 * kp.sa_cv1 * kp.sa_cv1;
    const double kar2Inv = 1.0 / (kp.sa_k * kp.sa_k);
    const double cw3_2 = kp.sa_cw3 * kp.sa_cw3;
    const double cw36 = cw3_2 * cw3_2 * cw3_2;
    const adf_real8 cb3Inv = 1.0 / kp.sa_cb3;
    const double nu = b.rlv[c] / rho;
    const double d2 = b.d2wall[c];
    const double dist2Inv = 1.0 / (d2 * d2);
    const double chi = nut / nu;
    const double chi2 = chi * chi, chi3 = chi * chi2;
    const double fv1 = chi3 / (chi3 + cv13);
    const double fv2 = 1.0 - chi / (1.0 + chi * fv1);
    const double ft2 = kp.useft2SA ? kp.sa_ct3 * exp(-kp.sa_ct4 * chi2) : 0.0;
    double sst = ss + nut * fv2 * kar2Inv * dist2Inv;
    if (kp.useRotationSA) sst = sst + kp.sa_crot * fmin(0.0, sqrt(2.0 * strainMag2));
    sst = fmax(sst, 1.e-10);
    double rr = nut * kar2Inv * dist2Inv / sst;
    rr = fmin(rr, 10.0);
    const double rr2 = rr * rr;
    const double gg = rr + kp.sa_cw2 * (rr2 * rr2 * rr2 - rr);
    const double gg2 = gg * gg;
    const double gg6 = gg2 * gg2 * gg2;
    const double termFw = pow((1.0 + cw36) / (gg6 + cw36), 1.0 / 6.0);
    const double fwSa = gg * termFw;
    const double term1 = kp.sa_cb1 * (1.0 - ft2) * ss;
    const double term2 = dist2Inv * (kar2Inv * kp.sa_cb1 * ((1.0 - ft2) * fv2 + ft2) - kp.sa_cw1 * fwSa);
    double dvt = (term1 + term2 * nut) * nut;
    double qq = 0.0;
    if (SOLVE) {
        // -d(source)/d(nuTilde), clipped at zero (sa.F90:306-332)
        const double t1 = chi3 + cv13;
        const double dfv1 = 3.0 * chi2 * cv13 / (t1 * t1);
        const double t2 = 1.0 + chi * fv1;
        const double dfv2 = (chi2 * dfv1 - 1.0) / (nu * (t2 * t2));
        const double dft2 = -2.0 * kp.sa_ct4 * chi * ft2 / nu;
        const double drr = (1.0 - rr * (fv2 + nut * dfv2)) * kar2Inv * dist2Inv / sst;
        const double dgg = (1.0 - kp.sa_cw2 + 6.0 * kp.sa_cw2 * (rr2 * rr2 * rr)) * drr;
        const double dfw = (cw36 / (gg6 + cw36)) * termFw * dgg;
        qq = -2.0 * term2 * nut - dist2Inv * nut * nut * (kp.sa_cb1 * kar2Inv * (dfv2 - ft2 * dfv2 - fv2 * dft2 + dft2) - kp.sa_cw1 * dfw);
        qq = fmax(qq, 0.0);
    }

    // ---- advection, sweeps k, j, i (turbUtils.F90:886, 1118, 1349)
    const bool secondOrd = (kp.orderTurb == 2) && kp.groundLevelIsOne;
    double uu, c1m, c1p;
    // implicit boundary treatment (only where it adds diagonal dominance): max(bmt, 0) of the face behind a
    // boundary cell, zero elsewhere (turbUtils.F90:986-1004,1074-1092; sa.F90:452-468)
    double bmK1 = 0.0, bmK2 = 0.0, bmJ1 = 0.0, bmJ2 = 0.0, bmI1 = 0.0, bmI2 = 0.0;
    if (SOLVE && b.bmt[0]) {
        if (i == 2) bmI1 = fmax(b.bmt[0][(j - 1) + (long)b.je * (k - 1)], 0.0);
        if (i == b.il) bmI2 = fmax(b.bmt[1][(j - 1) + (long)b.je * (k - 1)], 0.0);
        if (j == 2) bmJ1 = fmax(b.bmt[2][(i - 1) + (long)b.ie * (k - 1)], 0.0);
        if (j == b.jl) bmJ2 = fmax(b.bmt[3][(i - 1) + (long)b.ie * (k - 1)], 0.0);
        if (k == 2) bmK1 = fmax(b.bmt[4][(i - 1) + (long)b.ie * (j - 1)], 0.0);
        if (k == b.kl) bmK2 = fmax(b.bmt[5][(i - 1) + (long)b.ie * (j - 1)], 0.0);
    }
    // central jacobian of the advection: +uu (uu > 0) or -uu, plus the boundary part (turbUtils.F90:972-1004,1060-1092)
    dvt += sa_advect(dk, vol0, u, v, w, secondOrd, &uu); qq += fabs(uu) + ((uu > 0.0) ? uu * bmK1 : -uu * bmK2);
    dvt += sa_advect(dj, vol0, u, v, w, secondOrd, &uu); qq += fabs(uu) + ((uu > 0.0) ? uu * bmJ1 : -uu * bmJ2);
    dvt += sa_advect(di, vol0, u, v, w, secondOrd, &uu); qq += fabs(uu) + ((uu > 0.0) ? uu * bmI1 : -uu * bmI2);

    // ---- diffusion, sweeps k, j, i (sa.F90:371, 473, 572); boundary cells: c1 - b1 max(bmt1,0) at index 2,
    //      else c1 - d1 max(bmt2,0) at the last index
    dvt += sa_diffuse(dk, vol0, nu, kp.sa_cb2, cb3Inv, &c1m, &c1p); qq += c1m + c1p + ((k == 2) ? c1m * bmK1 : c1p * bmK2);
    dvt += sa_diffuse(dj, vol0, nu, kp.sa_cb2, cb3Inv, &c1m, &c1p); qq += c1m + c1p + ((j == 2) ? c1m * bmJ1 : c1p * bmJ2);
    dvt += sa_diffuse(di, vol0, nu, kp.sa_cb2, cb3Inv, &c1m, &c1p); qq += c1m + c1p + ((i == 2) ? c1m * bmI1 : c1p * bmI2);

    // ---- scale (sa.F90:702-706)
    b.dw[c + 5 * nb] = -b.volRef[c] * dvt * flg_blank(b.flags[c]);
    if (SOLVE) {
        b.scratch[c] = dvt;
        // saSolve scales the central jacobian by 1 + (1-alfa)/alfa for implicit relaxation (sa.F90:830-836)
        b.scratch[c + nb] = kp.sa_qqFactor * qq;
    }
}

// One direction of saSolve (sa.F90:858-1240): per line  bb = (-c1m - max(uu,0)) rblank,
// dd = (-c1p + min(uu,0)) rblank, cc = qq, ff = rhs rblank; elimination from the END of
// the line towards index 2, then forward substitution.  scratch: 0 rhs/solution, 1 qq,
// 2 modified cc, 3 bb.  LAST: the k sweep, followed by the update of nuTilde and rev.
// PRE: bb and dd of the direction were left in scratch (3, 4) j / (5, 6) i / (7, 8) k by k_sa_march<true>; otherwise they are formed here
template <int DIR, bool PRE = false>
__global__ __launch_bounds__(64) void k_sa_sweep(const BlkView* __restrict__ tab, KParams kp, int slot0)
{
    const BlkView& b = tab[slot0 + blockIdx.z + 1];
    const int a = blockIdx.x * 64 + threadIdx.x + 2;
    const int bbi = blockIdx.y + 2;
    int n, amax, bmax;
    long c0, s;
    const adf_real8* sN;
    if (DIR == 0) { amax = b.jl; bmax = b.kl; n = b.nx; c0 = b.idx(2, a, bbi); s = 1; sN = b.sI; }
    else if (DIR == 1) { amax = b.il; bmax = b.kl; n = b.ny; c0 = b.idx(a, 2, bbi); s = b.ldi; sN = b.sJ; }
    else { amax = b.il; bmax = b.jl; n = b.nz; c0 = b.idx(a, bbi, 2); s = b.ldk; sN = b.sK; }
    if (b.nx == 0 || a > amax || bbi > bmax) return;
    const long nb = b.nbox;
    const adf_real8 cb3Inv = 1.0 / kp.sa_cb3;
    double* rhs = b.scratch;
    double* qqA = b.scratch + nb;
    double* ccA = b.scratch + 2 * nb;
    double* bbA = b.scratch + (PRE ? (DIR == 1 ? 3 : (DIR == 0 ? 5 : 7)) : 3) * nb;
    const double* ddA = b.scratch + (DIR == 1 ? 4 : (DIR == 0 ? 6 : 8)) * nb;
    // backward elimination: m = n-1 (index jl) down to 0 (index 2)
    double ccN = 1.0, bbN = 0.0, ffN = 0.0;   // values of row m+1
    for (int m = n - 1; m >= 0; --m) {
        const long c = c0 + m * s;
        const double rblank = flg_blank(b.flags[c]);
        double bb, dd;
        if (PRE) { bb = bbA[c]; dd = ddA[c]; }
        else {
            SaDir d;
            load_dir(b, c, s, sN, d, DIR);
            const adf_real8 vol0 = b.vol[c];
            const double nu = b.rlv[c] / b.w[c];
            double c1m, c1p, uu;
            (void)sa_diffuse(d, vol0, nu, kp.sa_cb2, cb3Inv, &c1m, &c1p);
            (void)sa_advect(d, vol0, b.w[c + nb], b.w[c + 2 * nb], b.w[c + 3 * nb], false, &uu);
            const double um = (uu < 0.0) ? uu : 0.0, up = (uu > 0.0) ? uu : 0.0;
            bb = (-c1m - up) * rblank;
            dd = (-c1p + um) * rblank;
        }
        double cc = qqA[c];
        double ff = rhs[c] * rblank;
        if (m < n - 1) {
            const double f = dd / ccN;
            cc = cc - f * bbN;
            ff = ff - f * ffN;
        }
        ccA[c] = cc;
        if (!PRE) bbA[c] = bb;
        rhs[c] = ff;
        ccN = cc; bbN = bb; ffN = ff;
    }
    // forward substitution
    double fprev = 0.0;
    for (int m = 0; m < n; ++m) {
        const long c = c0 + m * s;
        double ff = rhs[c];
        if (m > 0) ff = ff - bbA[c] * fprev;
        ff = ff / ccA[c];
        fprev = ff;
        if (DIR == 2) {
            // last sweep: update nuTilde (explicit relaxation factor) and the eddy viscosity
            // (sa.F90:1248-1262, saEddyViscosity turbUtils.F90:657-720)
            double nut = b.w[c + 5 * nb] + kp.sa_updFactor * ff;
            nut = fmax(nut, 0.0);
            b.w[c + 5 * nb] = nut;
            rhs[c] = ff;
            const double cv13 = kp.sa_cv1 * kp.sa_cv1 * kp.sa_cv1;
            const double rnuSA = nut * b.w[c];
            const double chi = rnuSA / b.rlv[c];
            const double chi3 = chi * chi * chi;
            b.rev[c] = chi3 / (chi3 + cv13) * rnuSA;
        } else {
            rhs[c] = ff * qqA[c];   // right-hand side of the next direction
        }
    }
}

// ---------------------------------------------------------------------------
// i-direction sweep of saSolve in coalesced pieces (same arithmetic as k_sa_sweep<0>; see the
// D-ADI counterpart in kernels_smooth.hip):
//   k_sa_rows_i   pointwise, lanes over i: bb (scratch 3), dd (scratch 4), ff = rhs * rblank (scratch 0)
//   k_sa_solve_i  one thread per line: elimination from the end of the line, forward substitution,
//                 rhs of the next direction = solution * qq; 64-line x 16-cell LDS tiles
// ---------------------------------------------------------------------------
#define SI_SH 3
#define SI_CH (1 << SI_SH)
#define SI_LD (SI_CH + 1)
#define SI_LPA (64 >> SI_SH)

__device__ __forceinline__ void sa_tile_load(const BlkView& b, const double* __restrict__ arr, double* __restrict__ tile, int j0, int k,
                                             int i0, int lane)
{
    const int sub = lane >> SI_SH, col = lane & (SI_CH - 1), i = i0 + col;
#pragma unroll
    for (int q = 0; q < SI_CH; ++q) {
        const int r = SI_LPA * q + sub;
        tile[r * SI_LD + col] = (j0 + r <= b.jl && i <= b.il) ? arr[b.idx(i, j0 + r, k)] : 1.0;
    }
}

__device__ __forceinline__ void sa_tile_store(const BlkView& b, double* __restrict__ arr, const double* __restrict__ tile, int j0, int k,
                                              int i0, int lane)
{
    const int sub = lane >> SI_SH, col = lane & (SI_CH - 1), i = i0 + col;
#pragma unroll
    for (int q = 0; q < SI_CH; ++q) {
        const int r = SI_LPA * q + sub;
        if (j0 + r <= b.jl && i <= b.il) arr[b.idx(i, j0 + r, k)] = tile[r * SI_LD + col];
    }
}

__global__ __launch_bounds__(SA_BX* SA_BY) void k_sa_rows_i(const BlkView* __restrict__ tab, int nzb, KParams kp)
{
    const BlkView& b = tab[blockIdx.z / nzb + 1];
    const int i = blockIdx.x * SA_BX + threadIdx.x + 2;
    const int j = blockIdx.y * SA_BY + threadIdx.y + 2;
    const int k = blockIdx.z % nzb + 2;
    if (i > b.il || j > b.jl || k > b.kl) return;
    const long c = b.idx(i, j, k), nb = b.nbox;
    const adf_real8 cb3Inv = 1.0 / kp.sa_cb3;
    SaDir d;
    load_dir(b, c, 1, b.sI, d, 0);
    const adf_real8 vol0 = b.vol[c];
    const double nu = b.rlv[c] / b.w[c];
    double c1m, c1p, uu;
    (void)sa_diffuse(d, vol0, nu, kp.sa_cb2, cb3Inv, &c1m, &c1p);
    (void)sa_advect(d, vol0, b.w[c + nb], b.w[c + 2 * nb], b.w[c + 3 * nb], false, &uu);
    const double rblank = flg_blank(b.flags[c]);
    const double um = (uu < 0.0) ? uu : 0.0, up = (uu > 0.0) ? uu : 0.0;
    b.scratch[c + 3 * nb] = (-c1m - up) * rblank;
    b.scratch[c + 4 * nb] = (-c1p + um) * rblank;
    b.scratch[c] = b.scratch[c] * rblank;
}

// register form of the tile transposition (see k_dadi_solve_i, kernels_smooth.hip): coalesced global loads into registers,
// one LDS tile per workgroup (one wave), all arrays of a chunk in flight before the first transposition
__device__ __forceinline__ void sa_tile_fetch(const BlkView& b, const double* __restrict__ arr, int j0, int k, int i0, int lane,
                                              double raw[SI_CH])
{
    const int sub = lane >> SI_SH, col = lane & (SI_CH - 1), i = i0 + col;
#pragma unroll
    for (int q = 0; q < SI_CH; ++q) {
        const int r = SI_LPA * q + sub;
        raw[q] = (j0 + r <= b.jl && i <= b.il) ? arr[b.idx(i, j0 + r, k)] : 1.0;
    }
}

__device__ __forceinline__ void sa_tile_to_line(double* __restrict__ tile, int lane, const double raw[SI_CH], double v[SI_CH])
{
    const int sub = lane >> SI_SH, col = lane & (SI_CH - 1);
#pragma unroll
    for (int q = 0; q < SI_CH; ++q) tile[(SI_LPA * q + sub) * SI_LD + col] = raw[q];
    __syncthreads();
#pragma unroll
    for (int m = 0; m < SI_CH; ++m) v[m] = tile[lane * SI_LD + m];
    __syncthreads();
}

__device__ __forceinline__ void sa_line_to_global(const BlkView& b, double* __restrict__ arr, double* __restrict__ tile, int j0, int k,
                                                  int i0, int lane, const double v[SI_CH])
{
#pragma unroll
    for (int m = 0; m < SI_CH; ++m) tile[lane * SI_LD + m] = v[m];
    __syncthreads();
    sa_tile_store(b, arr, tile, j0, k, i0, lane);
    __syncthreads();
}

// PRE: bb, dd of the i direction in scratch 5, 6 (k_sa_march<true>) and the right-hand side not yet blanked; otherwise in 3, 4 with the
// blanked right-hand side (k_sa_rows_i)
template <bool PRE>
__global__ __launch_bounds__(64) void k_sa_solve_i(const BlkView* __restrict__ tab)
{
    __shared__ double tile[64 * SI_LD];
    const BlkView& b = tab[blockIdx.z + 1];
    const int lane = threadIdx.x;
    const int j0 = blockIdx.x * 64 + 2, k = blockIdx.y + 2;
    const int n = b.nx;
    if (b.nx == 0 || j0 > b.jl || k > b.kl) return;
    const long nb = b.nbox;
    double* __restrict__ rhs = b.scratch;
    const double* __restrict__ qqA = b.scratch + nb;
    double* __restrict__ ccA = b.scratch + 2 * nb;
    const double* __restrict__ bbA = b.scratch + (PRE ? 5 : 3) * nb;
    const double* __restrict__ ddA = b.scratch + (PRE ? 6 : 4) * nb;
    const int nch = (n + SI_CH - 1) / SI_CH;
    // elimination from the end of the line: chunks right to left
    double ccN = 1.0, bbN = 0.0, ffN = 0.0;
    for (int ch = nch - 1; ch >= 0; --ch) {
        const int i0 = 2 + ch * SI_CH;
        double rb[SI_CH], rc[SI_CH], rd[SI_CH], rf[SI_CH], vb[SI_CH], vc[SI_CH], vd[SI_CH], vf[SI_CH];
        sa_tile_fetch(b, bbA, j0, k, i0, lane, rb); sa_tile_fetch(b, qqA, j0, k, i0, lane, rc);
        sa_tile_fetch(b, ddA, j0, k, i0, lane, rd); sa_tile_fetch(b, rhs, j0, k, i0, lane, rf);
        sa_tile_to_line(tile, lane, rb, vb); sa_tile_to_line(tile, lane, rc, vc);
        sa_tile_to_line(tile, lane, rd, vd); sa_tile_to_line(tile, lane, rf, vf);
        int mTop = n - 1 - ch * SI_CH;
        if (mTop > SI_CH - 1) mTop = SI_CH - 1;
#pragma unroll
        for (int m = SI_CH - 1; m >= 0; --m) {
            if (m <= mTop) {
                double cc = vc[m], ff = vf[m];
                const double bb = vb[m];
                if (ch * SI_CH + m < n - 1) {
                    const double f = vd[m] / ccN;
                    cc = cc - f * bbN;
                    ff = ff - f * ffN;
                }
                vc[m] = cc; vf[m] = ff;
                ccN = cc; bbN = bb; ffN = ff;
            }
        }
        sa_line_to_global(b, ccA, tile, j0, k, i0, lane, vc);
        sa_line_to_global(b, rhs, tile, j0, k, i0, lane, vf);
    }
    // forward substitution, then the right-hand side of the next direction
    double fprev = 0.0;
    for (int ch = 0; ch < nch; ++ch) {
        const int i0 = 2 + ch * SI_CH;
        double rb[SI_CH], rc[SI_CH], rd[SI_CH], rf[SI_CH], vb[SI_CH], vc[SI_CH], vd[SI_CH], vf[SI_CH];
        sa_tile_fetch(b, bbA, j0, k, i0, lane, rb); sa_tile_fetch(b, ccA, j0, k, i0, lane, rc);
        sa_tile_fetch(b, qqA, j0, k, i0, lane, rd); sa_tile_fetch(b, rhs, j0, k, i0, lane, rf);
        sa_tile_to_line(tile, lane, rb, vb); sa_tile_to_line(tile, lane, rc, vc);
        sa_tile_to_line(tile, lane, rd, vd); sa_tile_to_line(tile, lane, rf, vf);
        const int mEnd = (n - ch * SI_CH < SI_CH) ? n - ch * SI_CH : SI_CH;
#pragma unroll
        for (int m = 0; m < SI_CH; ++m) {
            if (m < mEnd) {
                double ff = vf[m];
                if (ch * SI_CH + m > 0) ff = ff - vb[m] * fprev;
                ff = ff / vc[m];
                fprev = ff;
                vf[m] = ff * vd[m];
            }
        }
        sa_line_to_global(b, rhs, tile, j0, k, i0, lane, vf);
    }
}


#ifndef ADF_AD_BUILD
// Round 4: the i direction of the DDADI solve as parallel cyclic reduction along the lanes (the scheme of k_dadi_i_pcr,
// kernels_smooth.hip): a workgroup of NW wavefronts holds one i line, every lane one row  bb x(i-1) + qq x(i) + dd x(i+1) = rhs,
// normalised by its diagonal; ceil(log2 nx) reduction steps through LDS, then the right-hand side of the k sweep rhs = x qq.
// One coalesced read of (bb, qq, dd, rhs) and one write of rhs per cell instead of the transposing tiles (counted 133 B per cell).
#define SP_JL 8
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_sa_i_pcr(const BlkView* __restrict__ tab)
{
    constexpr int T = 64 * NW;
    __shared__ double P[2 * 3 * T];
    const BlkView& b = tab[blockIdx.z + 1];
    const int t = threadIdx.x, n = b.nx;
    const int k = blockIdx.y + 2;
    const int j0 = blockIdx.x * SP_JL + 2;
    if (b.nx == 0 || k > b.kl || j0 > b.jl || n > T) return;        // uniform per workgroup
    const bool act = t < n;
    const int tc = act ? t : n - 1;
    const long nb = b.nbox;
    double* __restrict__ rhs = b.scratch;
    const double* __restrict__ qqA = b.scratch + nb;
    const double* __restrict__ bbA = b.scratch + 5 * nb;
    const double* __restrict__ ddA = b.scratch + 6 * nb;
    for (int jl_ = 0; jl_ < SP_JL; ++jl_) {
        const int j = j0 + jl_;
        if (j > b.jl) break;                                         // uniform
        const long c = b.idx(2 + tc, j, k);
        const double qq = qqA[c];
        const double inv = rcp_nr(qq);
        double a = (act && t > 0) ? bbA[c] * inv : 0.0;
        double cc = (act && t < n - 1) ? ddA[c] * inv : 0.0;
        double d = act ? rhs[c] * inv : 0.0;
        int cur_ = 0;
        for (int st = 1; st < n; st <<= 1) {
            double* __restrict__ Q = P + cur_ * 3 * T;
            Q[t] = a; Q[T + t] = cc; Q[2 * T + t] = d;
            __syncthreads();
            const bool lo = t >= st, hi = t + st < n;
            const int im = lo ? t - st : t, ip = hi ? t + st : t;
            const double am = lo ? Q[im] : 0.0, cm = lo ? Q[T + im] : 0.0, dm = lo ? Q[2 * T + im] : 0.0;
            const double ap = hi ? Q[ip] : 0.0, cp = hi ? Q[T + ip] : 0.0, dp = hi ? Q[2 * T + ip] : 0.0;
            const double al = -a, ga = -cc;
            const double iv = rcp_nr(1.0 + al * cm + ga * ap);
            d = (d + al * dm + ga * dp) * iv;
            a = al * am * iv;
            cc = ga * cp * iv;
            cur_ ^= 1;
        }
        if (act) rhs[c] = d * qq;
        __syncthreads();
    }
}

template <int NW>
static void launch_sa_i_pcr(const BlkView* tab, int nslots, int ny, int nz, hipStream_t s)
{
    hipLaunchKernelGGL((k_sa_i_pcr<NW>), dim3((ny + SP_JL - 1) / SP_JL, nz, nslots), dim3(64 * NW, 1, 1), 0, s, tab);
}
#endif

// marchRes: residual, right-hand side and central jacobian were left by the k-marching kernel (launch_sa_march, blocks at rest);
// otherwise the gather kernel forms them here
void launch_sa_solve_level(const BlkView* tab, int nslots, int nx, int ny, int nz, const KParams& kp, hipStream_t s, bool marchRes)
{
    LEVEL_SPLIT(nslots, nz + 4, launch_sa_solve_level(tab + s0_, n_, nx, ny, nz, kp, s, marchRes));
    if (nslots <= 0) return;
    dim3 blk(SA_BX, SA_BY, 1);
    dim3 grd((nx + SA_BX - 1) / SA_BX, (ny + SA_BY - 1) / SA_BY, nz * nslots);
    dim3 l64(64, 1, 1);
    // sweep order of the reference: j, i, k
    if (marchRes) {
        // the marching kernel left the off-diagonals of all three directions beside the right-hand side and the central jacobian
        hipLaunchKernelGGL((k_sa_sweep<1, true>), dim3((nx + 63) / 64, nz, nslots), l64, 0, s, tab, kp, 0);
#ifndef ADF_AD_BUILD
        if (g_dadi_pcr && nx <= 256) {
            if (nx <= 64) launch_sa_i_pcr<1>(tab, nslots, ny, nz, s);
            else if (nx <= 128) launch_sa_i_pcr<2>(tab, nslots, ny, nz, s);
            else if (nx <= 192) launch_sa_i_pcr<3>(tab, nslots, ny, nz, s);
            else launch_sa_i_pcr<4>(tab, nslots, ny, nz, s);
        } else
#endif
            hipLaunchKernelGGL((k_sa_solve_i<true>), dim3((ny + 63) / 64, nz, nslots), l64, 0, s, tab);
        hipLaunchKernelGGL((k_sa_sweep<2, true>), dim3((nx + 63) / 64, ny, nslots), l64, 0, s, tab, kp, 0);
        return;
    }
    hipLaunchKernelGGL((k_sa_residual<true>), grd, blk, 0, s, tab, nz, kp);
    hipLaunchKernelGGL((k_sa_sweep<1>), dim3((nx + 63) / 64, nz, nslots), l64, 0, s, tab, kp, 0);
    hipLaunchKernelGGL(k_sa_rows_i, grd, blk, 0, s, tab, nz, kp);
    hipLaunchKernelGGL((k_sa_solve_i<false>), dim3((ny + 63) / 64, nz, nslots), l64, 0, s, tab);
    hipLaunchKernelGGL((k_sa_sweep<2>), dim3((nx + 63) / 64, ny, nslots), l64, 0, s, tab, kp, 0);
}

void launch_sa_residual_level(const BlkView* tab, int nslots, int nx, int ny, int nz, const KParams& kp, hipStream_t s)
{
    LEVEL_SPLIT(nslots, nz + 4, launch_sa_residual_level(tab + s0_, n_, nx, ny, nz, kp, s));
    if (nslots <= 0) return;
    dim3 blk(SA_BX, SA_BY, 1);
    dim3 grd((nx + SA_BX - 1) / SA_BX, (ny + SA_BY - 1) / SA_BY, nz * nslots);
    hipLaunchKernelGGL((k_sa_residual<false>), grd, blk, 0, s, tab, nz, kp);
}
