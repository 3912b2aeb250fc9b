// Inviscid residual: central Euler flux + artificial dissipation, fused, as a
// cell-centred GATHER (each thread evaluates the six faces of its cell; a face
// flux is a pure function of the two/four cells around it so both neighbours
// obtain bit-identical values and the scheme stays conservative without
// atomics or the reference's scatter loops).
//
// Reference semantics:
//   central flux        fluxes::inviscidCentralFlux     src/solver/fluxes.F90:4-401
//   scalar JST          fluxes::inviscidDissFluxScalar  src/solver/fluxes.F90:1049-1436
//   matrix JST          fluxes::inviscidDissFluxMatrix  src/solver/fluxes.F90:403-1047
//   Roe upwind (MUSCL)  fluxes::inviscidUpwindFlux      src/solver/fluxes.F90:1438-2532
//   final sum           residual_block                  src/solver/residuals.F90:334-344
//
// Roofline: HBM (175 B/cell Euler).  No MFMA: 7/13-point stencil.
#include "internal.h"
#include "flux_faces.h"

#define IV_BX 64
#define IV_BY 4

// one index direction for the cell at line position 2: central flux through
// both faces + the selected dissipation
template <int SCHEME, bool VISC>
__device__ __forceinline__ void dir_flux(const BlkView& b, const KParams& kp, long c, long s, const adf_real8* __restrict__ sN,
                                         const double* __restrict__ rad, int porM, int porP, double sslim,
                                         double fis2, double fis4, bool doDiss, int lim, double dwc[5], double fwd[5],
                                         const adf_real8* __restrict__ sF)
{
    Line L;
    load_line(b, c, s, L);
    const long nb = b.nbox;
    // minus face: normal stored at the left cell c-s ; plus face at c
    const double mx = sN[c - s], my = sN[c - s + nb], mz = sN[c - s + 2 * nb];
    const double px = sN[c], py = sN[c + nb], pz = sN[c + 2 * nb];
    // grid velocity through the two faces (sFaceI/J/K of a moving block)
    double sfM = 0.0, sfP = 0.0;
    if (sF) { sfM = sF[c - s]; sfP = sF[c]; }     // uniform branch
    central_face(L, 1, mx, my, mz, porM, -1.0, dwc, sfM);
    central_face(L, 2, px, py, pz, porP, +1.0, dwc, sfP);
    if (!doDiss) return;
    if (SCHEME == ADFLOW_DISS_SCALAR && !kp.fineGrid) {
        // coarse multigrid levels: first-order scalar dissipation
        // (fluxes::inviscidDissFluxScalarCoarse, fluxes.F90:4977-5203): fs = dis0 (W_R - W_L)
        const double fis0 = kp.rFil * kp.vis2Coarse;
        const double r0 = rad[c];
        const double d0M = fis0 * (porM == ADF_POR_NORMAL ? 0.5 : 0.0) * (rad[c - s] + r0);
        const double d0P = fis0 * (porP == ADF_POR_NORMAL ? 0.5 : 0.0) * (r0 + rad[c + s]);
        double Wm[5], W0[5], Wp[5];
        Wm[0] = L.rho[1]; Wm[1] = L.rho[1] * L.u[1]; Wm[2] = L.rho[1] * L.v[1]; Wm[3] = L.rho[1] * L.w[1]; Wm[4] = L.e[1] + L.p[1];
        W0[0] = L.rho[2]; W0[1] = L.rho[2] * L.u[2]; W0[2] = L.rho[2] * L.v[2]; W0[3] = L.rho[2] * L.w[2]; W0[4] = L.e[2] + L.p[2];
        Wp[0] = L.rho[3]; Wp[1] = L.rho[3] * L.u[3]; Wp[2] = L.rho[3] * L.v[3]; Wp[3] = L.rho[3] * L.w[3]; Wp[4] = L.e[3] + L.p[3];
#pragma unroll
        for (int l = 0; l < 5; ++l) fwd[l] += d0M * (W0[l] - Wm[l]) - d0P * (Wp[l] - W0[l]);
    } else if (SCHEME == ADFLOW_DISS_SCALAR) {
        double ssv[5];
        const bool approx = kp.dissApprox != 0;
        if (VISC || approx) {      // approx: the frozen sensor (adflow_gpu_reference_shock_sensor) for Euler too
#pragma unroll
            for (int m = 0; m < 5; ++m) ssv[m] = b.ss[c + (m - 2) * s];
        } else {
#pragma unroll
            for (int m = 0; m < 5; ++m) ssv[m] = L.p[m];
        }
        const double dm = jst_sensor(ssv[0], ssv[1], ssv[2], sslim);
        const double d0 = jst_sensor(ssv[1], ssv[2], ssv[3], sslim);
        const double dp = jst_sensor(ssv[2], ssv[3], ssv[4], sslim);
        const double r0 = rad[c];
        const double rrM = (porM == ADF_POR_NORMAL ? 0.5 : 0.0) * (rad[c - s] + r0);
        const double rrP = (porP == ADF_POR_NORMAL ? 0.5 : 0.0) * (r0 + rad[c + s]);
        jst_scalar_face(L, 1, rrM, dm, d0, fis2, fis4, +1.0, fwd, approx, kp.sigma);   // cell is the right cell
        jst_scalar_face(L, 2, rrP, d0, dp, fis2, fis4, -1.0, fwd, approx, kp.sigma);   // cell is the left cell
    } else {
        double gam[5];
        gam[0] = gam[4] = 0.0;
#pragma unroll
        for (int m = 1; m < 4; ++m) gam[m] = b.gamma[c + (m - 2) * s];
        if (SCHEME == ADFLOW_DISS_MATRIX && !kp.fineGrid) {
            const double fis0 = kp.rFil * kp.vis2Coarse;
            jst_matrix_face(L, gam, 1, mx, my, mz, porM, 0.0, 0.0, fis0, 0.0, +1.0, fwd, true, false, 0.0, sfM);
            jst_matrix_face(L, gam, 2, px, py, pz, porP, 0.0, 0.0, fis0, 0.0, -1.0, fwd, true, false, 0.0, sfP);
        } else if (SCHEME == ADFLOW_DISS_MATRIX) {
            const bool approx = kp.dissApprox != 0;
            double sv[5];
#pragma unroll
            for (int m = 0; m < 5; ++m) sv[m] = approx ? b.ss[c + (m - 2) * s] : L.p[m];   // frozen pressure when approx
            const double dm = mat_sensor(sv[0], sv[1], sv[2], sslim);
            const double d0 = mat_sensor(sv[1], sv[2], sv[3], sslim);
            const double dp = mat_sensor(sv[2], sv[3], sv[4], sslim);
            jst_matrix_face(L, gam, 1, mx, my, mz, porM, dm, d0, fis2, fis4, +1.0, fwd, false, approx, kp.sigma, sfM);
            jst_matrix_face(L, gam, 2, px, py, pz, porP, d0, dp, fis2, fis4, -1.0, fwd, false, approx, kp.sigma, sfP);
        } else {   // Roe upwind: fw(left) += flux, fw(right) -= flux
            roe_face(L, gam, 1, mx, my, mz, porM, lim, kp.kappaCoef, kp.rFil, kp.gammaConstant, -1.0, fwd, sfM);
            roe_face(L, gam, 2, px, py, pz, porP, lim, kp.kappaCoef, kp.rFil, kp.gammaConstant, +1.0, fwd, sfP);
        }
    }
}

// FINAL: dw = (init + central + fw) * iblank written; otherwise the viscous kernel completes the sum: with the persistent
// fw of the Runge-Kutta scheme (fwMode) dw = init + central and fw are stored, else only dw = init + central + fw.
#ifdef ADF_AD_BUILD
#define IV_MINWG(S) 1       // dual numbers: the whole register file (256 registers + 1.4 KB of scratch per lane with two workgroups per CU)
#else
#define IV_MINWG(S) ((S) == ADFLOW_UPWIND ? 2 : 1)
#endif
// LIMT >= 0: the limiter at compile time (the first-order Roe flux of the coarse levels and of the preconditioner matrix: no
// reconstruction code in the kernel -- in the dual-number build that is the difference between 460 B of scratch per lane and none)
template <int SCHEME, bool VISC, bool FINAL, int LIMT = -1>
__global__ __launch_bounds__(IV_BX* IV_BY, IV_MINWG(SCHEME)) void k_inviscid(const BlkView* __restrict__ tab, int nzb, KParams kp)
{
    // level-batched: blockIdx.z = block slot * nzb + plane
    const BlkView& b = tab[blockIdx.z / nzb + 1];
    const int i = blockIdx.x * IV_BX + threadIdx.x + 2;
    const int j = blockIdx.y * IV_BY + threadIdx.y + 2;
    const int k = blockIdx.z % nzb + 2;
    if (i > b.il || j > b.jl || k > b.kl) return;
    const long c = b.idx(i, j, k);
    const long nb = b.nbox;

    const uint8_t f0 = b.flags[c];
    const uint8_t fi = b.flags[c - 1], fj = b.flags[c - b.ldi], fk = b.flags[c - b.ldk];

    double sslim;
    if (SCHEME == ADFLOW_DISS_SCALAR && VISC)
        sslim = 0.001 * kp.pInfCorr / pow(kp.rhoInf, kp.gammaInf);
    else
        sslim = 0.001 * kp.pInfCorr;   // scalar-Euler sslim / matrix plim
    const double fis2 = kp.rFil * kp.vis2, fis4 = kp.rFil * kp.vis4;
    const bool doDiss = fabs(kp.rFil) >= 1.e-10;
    // limiter actually used: first order off the fine grid (fluxes.F90:1531-1538)
    const int lim = (LIMT >= 0) ? LIMT : ((kp.fineGrid && !kp.lumpedDiss) ? kp.limiter : ADFLOW_LIM_FIRST_ORDER);

    double dwc[5] = {0, 0, 0, 0, 0}, fwd[5] = {0, 0, 0, 0, 0};
    const adf_real8* sF = b.sFace;
#ifdef ADF_AD_BUILD
    // dual numbers: the three directions one after the other in ONE copy of the code (a loop the compiler must not unroll), so that
    // only one direction's line of states is live at a time
    {
        const long strd[3] = {1, b.ldi, b.ldk};
        const adf_real8* sNd[3] = {b.sI, b.sJ, b.sK};
        const double* radd[3] = {b.radI, b.radJ, b.radK};
        const int pM[3] = {flg_porI(fi), flg_porJ(fj), flg_porK(fk)}, pP[3] = {flg_porI(f0), flg_porJ(f0), flg_porK(f0)};
#pragma nounroll
        for (int d = 0; d < 3; ++d)
            dir_flux<SCHEME, VISC>(b, kp, c, strd[d], sNd[d], radd[d], pM[d], pP[d], sslim, fis2, fis4, doDiss, lim, dwc, fwd,
                                   sF ? sF + d * nb : nullptr);
    }
#else
    dir_flux<SCHEME, VISC>(b, kp, c, 1, b.sI, b.radI, flg_porI(fi), flg_porI(f0), sslim, fis2, fis4, doDiss, lim, dwc, fwd, sF);
    dir_flux<SCHEME, VISC>(b, kp, c, b.ldi, b.sJ, b.radJ, flg_porJ(fj), flg_porJ(f0), sslim, fis2, fis4, doDiss, lim, dwc, fwd,
                           sF ? sF + nb : nullptr);
    dir_flux<SCHEME, VISC>(b, kp, c, b.ldk, b.sK, b.radK, flg_porK(fk), flg_porK(f0), sslim, fis2, fis4, doDiss, lim, dwc, fwd,
                           sF ? sF + 2 * nb : nullptr);
#endif
    if (b.moving) {
        // rotational source of the momentum equations, steady mode: the equations are solved in the inertial frame
        // (inviscidCentralFlux, fluxes.F90:372-397)
        const double wwx = kp.timeRef * b.rot[0], wwy = kp.timeRef * b.rot[1], wwz = kp.timeRef * b.rot[2];
        const double rvol = b.w[c] * b.vol[c];
        const double u = b.w[c + nb], v = b.w[c + 2 * nb], w = b.w[c + 3 * nb];
        dwc[1] += rvol * (wwy * w - wwz * v);
        dwc[2] += rvol * (wwz * u - wwx * w);
        dwc[3] += rvol * (wwx * v - wwy * u);
    }

    const double blank = flg_blank(f0);
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        double fwn = fwd[l];
        if (kp.fwMode) {
            // persistent dissipation residual of the RK scheme: fw = sfil*fw + new
            // (when rFil == 0 the stored value is reused unchanged, fluxes.F90:1085)
            const double old = b.fw[c + l * nb];
            fwn = doDiss ? (kp.sfil * old + fwd[l]) : old;
        }
        double d = dwc[l];
        if (kp.coarseInit) d += b.wr[c + l * nb];
        if (FINAL) {
            if (kp.fwMode && doDiss) b.fw[c + l * nb] = fwn;
            b.dw[c + l * nb] = (d + fwn) * blank;
        } else if (kp.fwMode) {
            b.fw[c + l * nb] = fwn;
            b.dw[c + l * nb] = d;
        } else {
            b.dw[c + l * nb] = (d + fwn) * blank;      // fw not persistent: the viscous kernel adds its part to dw(2:5) and re-applies iblank
        }
    }
}

template <int SCHEME>
static void launch_scheme(const BlkView* b, int nzb, const KParams& kp, dim3 grd, dim3 blk, hipStream_t s)
{
    // the viscous kernel completes the sum unless rFil == 0 (viscousFlux returns
    // early, fluxes.F90:2585: the stored fw already holds the viscous part)
    const bool doDiss = fabs(kp.rFil) >= 1.e-10;
    if (SCHEME == ADFLOW_UPWIND && ((kp.fineGrid && !kp.lumpedDiss) ? kp.limiter : ADFLOW_LIM_FIRST_ORDER) == ADFLOW_LIM_FIRST_ORDER) {
        constexpr int FO = ADFLOW_LIM_FIRST_ORDER;
        if (kp.viscous) {
            if (doDiss) hipLaunchKernelGGL((k_inviscid<SCHEME, true, false, FO>), grd, blk, 0, s, b, nzb, kp);
            else hipLaunchKernelGGL((k_inviscid<SCHEME, true, true, FO>), grd, blk, 0, s, b, nzb, kp);
        } else {
            hipLaunchKernelGGL((k_inviscid<SCHEME, false, true, FO>), grd, blk, 0, s, b, nzb, kp);
        }
        return;
    }
    if (kp.viscous) {
        if (doDiss)
            hipLaunchKernelGGL((k_inviscid<SCHEME, true, false>), grd, blk, 0, s, b, nzb, kp);
        else
            hipLaunchKernelGGL((k_inviscid<SCHEME, true, true>), grd, blk, 0, s, b, nzb, kp);
    } else {
        hipLaunchKernelGGL((k_inviscid<SCHEME, false, true>), grd, blk, 0, s, b, nzb, kp);
    }
}

void launch_inviscid_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const KParams& kp, hipStream_t s)
{
    LEVEL_SPLIT(nslots, maxnz + 4, launch_inviscid_level(tab + s0_, n_, maxnx, maxny, maxnz, kp, s));
    if (nslots <= 0) return;
    dim3 blk(IV_BX, IV_BY, 1);
    dim3 grd((maxnx + IV_BX - 1) / IV_BX, (maxny + IV_BY - 1) / IV_BY, maxnz * nslots);
    switch (kp.spaceDiscr) {
    case ADFLOW_DISS_SCALAR: launch_scheme<ADFLOW_DISS_SCALAR>(tab, maxnz, kp, grd, blk, s); break;
    case ADFLOW_DISS_MATRIX: launch_scheme<ADFLOW_DISS_MATRIX>(tab, maxnz, kp, grd, blk, s); break;
    case ADFLOW_UPWIND: launch_scheme<ADFLOW_UPWIND>(tab, maxnz, kp, grd, blk, s); break;
    }
}

// ---------------------------------------------------------------------------
// initres (residuals.F90:427-955, steady branch) for an explicit variable range
__global__ void k_initres(BlkView b, int l0, int l1, int coarse)
{
    const int i = blockIdx.x * 64 + threadIdx.x + 2;
    const int j = blockIdx.y * 4 + threadIdx.y + 2;
    const int k = blockIdx.z + 2;
    if (i > b.il || j > b.jl) return;
    const long c = b.idx(i, j, k);
    for (int l = l0; l <= l1; ++l) {
        double v = 0.0;
        if (coarse && l < 5) v = b.wr[c + l * b.nbox];
        b.dw[c + l * b.nbox] = v;
    }
}

__global__ void k_initres_level(const BlkView* __restrict__ tab, int nzb, int l0, int l1, int coarse)
{
    const BlkView& b = tab[blockIdx.z / nzb + 1];
    const int i = blockIdx.x * 64 + threadIdx.x + 2;
    const int j = blockIdx.y * 4 + threadIdx.y + 2;
    const int k = blockIdx.z % nzb + 2;
    if (i > b.il || j > b.jl || k > b.kl) return;
    const long c = b.idx(i, j, k);
    for (int l = l0; l <= l1; ++l) {
        double v = 0.0;
        if (coarse && l < 5) v = b.wr[c + l * b.nbox];
        b.dw[c + l * b.nbox] = v;
    }
}

void launch_initres_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const KParams& kp, int l0, int l1, hipStream_t s)
{
    LEVEL_SPLIT(nslots, maxnz + 4, launch_initres_level(tab + s0_, n_, maxnx, maxny, maxnz, kp, l0, l1, s));
    if (nslots <= 0) return;
    hipLaunchKernelGGL(k_initres_level, dim3((maxnx + 63) / 64, (maxny + 3) / 4, maxnz * nslots), dim3(64, 4, 1), 0, s, tab, maxnz, l0, l1,
                       kp.coarseInit);
}

void launch_initres(const BlkView& b, const KParams& kp, int l0, int l1, hipStream_t s)
{
    dim3 blk(64, 4, 1);
    dim3 grd((b.nx + 63) / 64, (b.ny + 3) / 4, b.nz);
    hipLaunchKernelGGL(k_initres, grd, blk, 0, s, b, l0, l1, kp.coarseInit);
}
