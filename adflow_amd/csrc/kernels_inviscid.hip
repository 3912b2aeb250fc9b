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

#define IV_BX 64
#define IV_BY 4

struct Line {       // 5-point line of primitive data along one index direction
    double rho[5], u[5], v[5], w[5], e[5], p[5];
};

__device__ __forceinline__ void load_line(const BlkView& b, long c, long s, Line& L)
{
    const long nb = b.nbox;
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        const long q = c + (m - 2) * s;
        L.rho[m] = b.w[q];
        L.u[m] = b.w[q + nb];
        L.v[m] = b.w[q + 2 * nb];
        L.w[m] = b.w[q + 3 * nb];
        L.e[m] = b.w[q + 4 * nb];
        L.p[m] = b.p[q];
    }
}

// central flux through the face between line positions l (left) and l+1, with
// face normal (sx,sy,sz) and porosity code `por`; adds +F to acc when the cell
// is the left one (sign=+1) and -F when it is the right one (sign=-1).
// fluxes.F90:52-129
__device__ __forceinline__ void central_face(const Line& L, int l, double sx, double sy, double sz, int por,
                                             double sign, double acc[5])
{
    const int r = l + 1;
    double vnp = L.u[r] * sx + L.v[r] * sy + L.w[r] * sz;
    double vnm = L.u[l] * sx + L.v[l] * sy + L.w[l] * sz;
    double porVel = 1.0, porFlux = 0.5;
    if (por == ADF_POR_NOFLUX) porFlux = 0.0;
    if (por == ADF_POR_BOUND) {
        porVel = 0.0;
        vnp = 0.0;   // sFace == 0: steady, non-moving blocks
        vnm = 0.0;
    }
    porVel *= porFlux;
    const double qsp = vnp * porVel, qsm = vnm * porVel;
    const double rqsp = qsp * L.rho[r], rqsm = qsm * L.rho[l];
    const double pa = porFlux * (L.p[r] + L.p[l]);
    acc[0] += sign * (rqsp + rqsm);
    acc[1] += sign * (rqsp * L.u[r] + rqsm * L.u[l] + pa * sx);
    acc[2] += sign * (rqsp * L.v[r] + rqsm * L.v[l] + pa * sy);
    acc[3] += sign * (rqsp * L.w[r] + rqsm * L.w[l] + pa * sz);
    acc[4] += sign * (qsp * L.e[r] + qsm * L.e[l] + porFlux * (vnp * L.p[r] + vnm * L.p[l]));
}

// scalar JST dissipative flux through face (l | l+1); needs line entries
// l-1..l+2.  fw(right) += fs, fw(left) -= fs  (fluxes.F90:1204-1272)
__device__ __forceinline__ void jst_scalar_face(const Line& L, int l, double rrad, double dssL, double dssR,
                                                double fis2, double fis4, double sign, double acc[5])
{
    const int r = l + 1, ll = l - 1, rr = l + 2;
    const double dis2 = fis2 * rrad * fmin(0.25, fmax(dssL, dssR));
    const double dis4 = fmax(fis4 * rrad - dis2, 0.0);   // myDim, utils.F90:470-480
    double ddw, fs;
    ddw = L.rho[r] - L.rho[l];
    fs = dis2 * ddw - dis4 * (L.rho[rr] - L.rho[ll] - 3.0 * ddw);
    acc[0] += sign * fs;
    ddw = L.u[r] * L.rho[r] - L.u[l] * L.rho[l];
    fs = dis2 * ddw - dis4 * (L.u[rr] * L.rho[rr] - L.u[ll] * L.rho[ll] - 3.0 * ddw);
    acc[1] += sign * fs;
    ddw = L.v[r] * L.rho[r] - L.v[l] * L.rho[l];
    fs = dis2 * ddw - dis4 * (L.v[rr] * L.rho[rr] - L.v[ll] * L.rho[ll] - 3.0 * ddw);
    acc[2] += sign * fs;
    ddw = L.w[r] * L.rho[r] - L.w[l] * L.rho[l];
    fs = dis2 * ddw - dis4 * (L.w[rr] * L.rho[rr] - L.w[ll] * L.rho[ll] - 3.0 * ddw);
    acc[3] += sign * fs;
    ddw = (L.e[r] + L.p[r]) - (L.e[l] + L.p[l]);
    fs = dis2 * ddw - dis4 * ((L.e[rr] + L.p[rr]) - (L.e[ll] + L.p[ll]) - 3.0 * ddw);
    acc[4] += sign * fs;
}

__device__ __forceinline__ double jst_sensor(double sm, double s0, double sp, double sslim)
{
    return fabs((sp - 2.0 * s0 + sm) / (sp + 2.0 * s0 + sm + sslim));
}

// one index direction of the scalar scheme for the cell at line position 2
template <bool VISC>
__device__ __forceinline__ void dir_scalar(const BlkView& b, const KParams& kp, long c, long s, const double* __restrict__ sN,
                                           const double* __restrict__ rad, int porM, int porP, double sslim,
                                           double fis2, double fis4, bool doDiss, double dwc[5], double fwd[5])
{
    Line L;
    load_line(b, c, s, L);
    const long nb = b.nbox;
    // minus face: normal stored at the left cell c-s ; plus face at c
    const double mx = sN[c - s], my = sN[c - s + nb], mz = sN[c - s + 2 * nb];
    const double px = sN[c], py = sN[c + nb], pz = sN[c + 2 * nb];
    central_face(L, 1, mx, my, mz, porM, -1.0, dwc);
    central_face(L, 2, px, py, pz, porP, +1.0, dwc);
    if (doDiss) {
        double ssv[5];
        if (VISC) {
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
        jst_scalar_face(L, 1, rrM, dm, d0, fis2, fis4, +1.0, fwd);   // cell is the right cell
        jst_scalar_face(L, 2, rrP, d0, dp, fis2, fis4, -1.0, fwd);   // cell is the left cell
    }
}

// FINAL: dw = (init + central + fw) * iblank written; otherwise dw = init + central
// and fw stored for the viscous kernel to complete.
template <bool VISC, bool FINAL>
__global__ __launch_bounds__(IV_BX* IV_BY) void k_inviscid_scalar(BlkView b, KParams kp)
{
    const int i = blockIdx.x * IV_BX + threadIdx.x + 2;
    const int j = blockIdx.y * IV_BY + threadIdx.y + 2;
    const int k = blockIdx.z + 2;
    if (i > b.il || j > b.jl) return;
    const long c = b.idx(i, j, k);
    const long nb = b.nbox;

    const uint8_t f0 = b.flags[c];
    const uint8_t fi = b.flags[c - 1], fj = b.flags[c - b.ldi], fk = b.flags[c - b.ldk];

    double sslim;
    if (VISC)
        sslim = 0.001 * kp.pInfCorr / pow(kp.rhoInf, kp.gammaInf);
    else
        sslim = 0.001 * kp.pInfCorr;
    const double fis2 = kp.rFil * kp.vis2, fis4 = kp.rFil * kp.vis4;
    const bool doDiss = fabs(kp.rFil) >= 1.e-10;

    double dwc[5] = {0, 0, 0, 0, 0}, fwd[5] = {0, 0, 0, 0, 0};
    dir_scalar<VISC>(b, kp, c, 1, b.sI, b.radI, flg_porI(fi), flg_porI(f0), sslim, fis2, fis4, doDiss, dwc, fwd);
    dir_scalar<VISC>(b, kp, c, b.ldi, b.sJ, b.radJ, flg_porJ(fj), flg_porJ(f0), sslim, fis2, fis4, doDiss, dwc, fwd);
    dir_scalar<VISC>(b, kp, c, b.ldk, b.sK, b.radK, flg_porK(fk), flg_porK(f0), sslim, fis2, fis4, doDiss, dwc, fwd);

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
        } else {
            b.fw[c + l * nb] = fwn;
            b.dw[c + l * nb] = d;
        }
    }
}

void launch_inviscid(const BlkView& b, const KParams& kp, hipStream_t s)
{
    dim3 blk(IV_BX, IV_BY, 1);
    dim3 grd((b.nx + IV_BX - 1) / IV_BX, (b.ny + IV_BY - 1) / IV_BY, b.nz);
    const bool final = !kp.viscous;
    if (kp.spaceDiscr == ADFLOW_DISS_SCALAR) {
        if (kp.viscous) {
            if (final)
                hipLaunchKernelGGL((k_inviscid_scalar<true, true>), grd, blk, 0, s, b, kp);
            else
                hipLaunchKernelGGL((k_inviscid_scalar<true, false>), grd, blk, 0, s, b, kp);
        } else {
            hipLaunchKernelGGL((k_inviscid_scalar<false, true>), grd, blk, 0, s, b, kp);
        }
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

void launch_initres(const BlkView& b, const KParams& kp, int l0, int l1, hipStream_t s)
{
    dim3 blk(64, 4, 1);
    dim3 grd((b.nx + 63) / 64, (b.ny + 3) / 4, b.nz);
    hipLaunchKernelGGL(k_initres, grd, blk, 0, s, b, l0, l1, kp.coarseInit);
}
