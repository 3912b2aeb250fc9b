// Spalart-Allmaras cell arithmetic shared by the gather kernel (kernels_sa.hip) and the marching kernel that evaluates the SA
// residual next to the nodal gradients (kernels_viscous.hip).
//   saSource        src/turbulence/sa.F90:89-344
//   turbAdvection   src/turbulence/turbUtils.F90:828-1561
//   saViscous       src/turbulence/sa.F90:346-676
#ifndef ADFLOW_SA_CORE_H
#define ADFLOW_SA_CORE_H
#include "internal.h"

// minmod-limited fully-upwind (kappa=-1) difference, or first order
// (turbUtils.F90:917-958 for uu>0, :1007-1047 for uu<=0)
__device__ __forceinline__ double upwind_diff(bool secondOrd, bool positive, double wm2, double wm1, double w0, double wp1,
                                              double wp2)
{
    if (positive) {
        if (!secondOrd) return w0 - wm1;
        const double dwtm1 = wm1 - wm2, dwt = w0 - wm1, dwtp1 = wp1 - w0;
        double d = dwt;
        if (dwt * dwtp1 > 0.0) d += (fabs(dwt) < fabs(dwtp1)) ? 0.5 * dwt : 0.5 * dwtp1;
        if (dwt * dwtm1 > 0.0) d -= (fabs(dwt) < fabs(dwtm1)) ? 0.5 * dwt : 0.5 * dwtm1;
        return d;
    } else {
        if (!secondOrd) return wp1 - w0;
        const double dwtm1 = w0 - wm1, dwt = wp1 - w0, dwtp1 = wp2 - wp1;
        double d = dwt;
        if (dwt * dwtp1 > 0.0) d -= (fabs(dwt) < fabs(dwtp1)) ? 0.5 * dwt : 0.5 * dwtp1;
        if (dwt * dwtm1 > 0.0) d += (fabs(dwt) < fabs(dwtm1)) ? 0.5 * dwt : 0.5 * dwtm1;
        return d;
    }
}

struct SaDir {   // per-direction data of one cell
    adf_real8 sm[3], sp[3];   // normals of the minus / plus face
    adf_real8 volm, volp;     // volumes of the minus / plus neighbour
    double nt[5];             // nuTilde at -2..+2
    double num, nup;          // laminar kinematic viscosity of the minus / plus neighbour
    adf_real8 qsf;            // grid velocity of a moving block: sFace(minus face) + sFace(plus face), else 0
};

// dirc: 0, 1, 2 = i, j, k (component of b.sFace)
__device__ __forceinline__ void load_dir(const BlkView& b, long c, long s, const adf_real8* __restrict__ sN, SaDir& d, int dirc)
{
    const long nb = b.nbox;
    d.qsf = 0.0;
    if (b.sFace) d.qsf = b.sFace[c + dirc * nb] + b.sFace[c - s + dirc * nb];     // uniform branch
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        d.sm[m] = sN[c - s + m * nb];
        d.sp[m] = sN[c + m * nb];
    }
    d.volm = b.vol[c - s];
    d.volp = b.vol[c + s];
#pragma unroll
    for (int m = 0; m < 5; ++m) d.nt[m] = b.w[c + (m - 2) * s + 5 * nb];
    d.num = b.rlv[c - s] / b.w[c - s];
    d.nup = b.rlv[c + s] / b.w[c + s];
}

// advection in one direction (turbUtils.F90:886-1070)
__device__ __forceinline__ double sa_advect(const SaDir& d, adf_real8 vol0, double u, double v, double w, bool secondOrd,
                                            double* uuOut = nullptr)
{
    const adf_real8 voli = 0.5 * rcp_nr(vol0);
    const adf_real8 xa = (d.sp[0] + d.sm[0]) * voli, ya = (d.sp[1] + d.sm[1]) * voli, za = (d.sp[2] + d.sm[2]) * voli;
    const double uu = xa * u + ya * v + za * w - d.qsf * voli;       // qs = (sFace(m) + sFace(m-1)) voli, turbUtils.F90:906
    const double dwt = upwind_diff(secondOrd, uu > 0.0, d.nt[0], d.nt[1], d.nt[2], d.nt[3], d.nt[4]);
    if (uuOut) *uuOut = uu;
    return -uu * dwt;
}

// diffusion in one direction (sa.F90:385-450)
__device__ __forceinline__ double sa_diffuse(const SaDir& d, adf_real8 vol0, double nu, adf_real8 cb2, adf_real8 cb3Inv,
                                             double* c1mOut = nullptr, double* c1pOut = nullptr)
{
    const adf_real8 voli = rcp_nr(vol0);
    const adf_real8 volmi = 2.0 * rcp_nr(vol0 + d.volm), volpi = 2.0 * rcp_nr(vol0 + d.volp);
    const adf_real8 xm = d.sm[0] * volmi, ym = d.sm[1] * volmi, zm = d.sm[2] * volmi;
    const adf_real8 xp = d.sp[0] * volpi, yp = d.sp[1] * volpi, zp = d.sp[2] * volpi;
    const adf_real8 xa = 0.5 * (d.sp[0] + d.sm[0]) * voli, ya = 0.5 * (d.sp[1] + d.sm[1]) * voli,
                    za = 0.5 * (d.sp[2] + d.sm[2]) * voli;
    const adf_real8 ttm = xm * xa + ym * ya + zm * za;
    const adf_real8 ttp = xp * xa + yp * ya + zp * za;
    const double cnud = -cb2 * d.nt[2] * cb3Inv;
    const double cam = ttm * cnud, cap = ttp * cnud;
    const double nutm = 0.5 * (d.nt[1] + d.nt[2]), nutp = 0.5 * (d.nt[3] + d.nt[2]);
    const double num = 0.5 * (d.num + nu), nup = 0.5 * (d.nup + nu);
    const double cdm = (num + (1.0 + cb2) * nutm) * ttm * cb3Inv;
    const double cdp = (nup + (1.0 + cb2) * nutp) * ttp * cb3Inv;
    const double c1m = fmax(cdm + cam, 0.0), c1p = fmax(cdp + cap, 0.0);
    const double c10 = c1m + c1p;
    if (c1mOut) { *c1mOut = c1m; *c1pOut = c1p; }
    return c1m * d.nt[1] - c10 * d.nt[2] + c1p * d.nt[3];
}


// source term of the SA equation for one cell (sa.F90:133-300), WITHOUT the jacobian part of saSolve: gu[m][d] = twice the
// volume times the gradient of velocity component m (the Green-Gauss sum over the six faces), returns dvt (before advection /
// diffusion)
// qqOut (saSolve): -d(source)/d(nuTilde) clipped at zero, sa.F90:306-332
__device__ __forceinline__ double sa_source(const KParams& kp, const double gu[3][3], adf_real8 vol0, double nu, double nut, adf_real8 d2,
                                            double* qqOut = nullptr)
{
    const adf_real8 fact = 0.25 * rcp_nr(vol0);
    double ss, strainMag2 = 0.0;
    if (kp.turbProd == ADFLOW_TURBPROD_STRAIN) {
        const double sxx = 2.0 * fact * gu[0][0], syy = 2.0 * fact * gu[1][1], szz = 2.0 * fact * gu[2][2];
        const double sxy = fact * (gu[0][1] + gu[1][0]), sxz = fact * (gu[0][2] + gu[2][0]), syz = fact * (gu[1][2] + gu[2][1]);
        const double tr = sxx + syy + szz;
        const double div2 = (2.0 * (1.0 / 3.0)) * (tr * tr);
        strainMag2 = 2.0 * (sxy * sxy + sxz * sxz + syz * syz) + sxx * sxx + syy * syy + szz * szz;
        ss = fastsqrt(fmax(2.0 * strainMag2 - div2, 0.0));
    } else {
        const double vortx = 2.0 * fact * (gu[2][1] - gu[1][2]);   // wheel speed omega = 0 (non-rotating sections)
        const double vorty = 2.0 * fact * (gu[0][2] - gu[2][0]);
        const double vortz = 2.0 * fact * (gu[1][0] - gu[0][1]);
        ss = fastsqrt(vortx * vortx + vorty * vorty + vortz * vortz);
    }
    const adf_real8 cv13 = kp.sa_cv1 * kp.sa_cv1 * kp.sa_cv1;
    const adf_real8 kar2Inv = 1.0 / (kp.sa_k * kp.sa_k);
    const adf_real8 cw3_2 = kp.sa_cw3 * kp.sa_cw3;
    const adf_real8 cw36 = cw3_2 * cw3_2 * cw3_2;
    const adf_real8 dist2Inv = rcp_nr(d2 * d2);
    const double chi = nut * rcp_nr(nu);
    const double chi2 = chi * chi, chi3 = chi * chi2;
    const double fv1 = chi3 * rcp_nr(chi3 + cv13);
    const double fv2 = 1.0 - chi * rcp_nr(1.0 + chi * fv1);
    const double ft2 = kp.useft2SA ? kp.sa_ct3 * fast_exp_neg(-kp.sa_ct4 * chi2) : 0.0;
    double sst = ss + nut * fv2 * kar2Inv * dist2Inv;
    if (kp.useRotationSA) sst = sst + kp.sa_crot * fmin(0.0, fastsqrt(2.0 * strainMag2));
    sst = fmax(sst, 1.e-10);
    double rr = nut * kar2Inv * dist2Inv * rcp_nr(sst);
    rr = fmin(rr, 10.0);
    const double rr2 = rr * rr;
    const double gg = rr + kp.sa_cw2 * (rr2 * rr2 * rr2 - rr);
    const double gg2 = gg * gg;
    const double gg6 = gg2 * gg2 * gg2;
    const double termFw = fast_root6((1.0 + cw36) * rcp_nr(gg6 + cw36));
    const double fwSa = gg * termFw;
    const double term1 = kp.sa_cb1 * (1.0 - ft2) * ss;
    const double term2 = dist2Inv * (kar2Inv * kp.sa_cb1 * ((1.0 - ft2) * fv2 + ft2) - kp.sa_cw1 * fwSa);
    if (qqOut) {
        const double t1 = chi3 + cv13;
        const double dfv1 = 3.0 * chi2 * cv13 * rcp_nr(t1 * t1);
        const double t2 = 1.0 + chi * fv1;
        const double nuInv = rcp_nr(nu);
        const double dfv2 = (chi2 * dfv1 - 1.0) * nuInv * rcp_nr(t2 * t2);
        const double dft2 = -2.0 * kp.sa_ct4 * chi * ft2 * nuInv;
        const double drr = (1.0 - rr * (fv2 + nut * dfv2)) * kar2Inv * dist2Inv * rcp_nr(sst);
        const double dgg = (1.0 - kp.sa_cw2 + 6.0 * kp.sa_cw2 * (rr2 * rr2 * rr)) * drr;
        const double dfw = (cw36 * rcp_nr(gg6 + cw36)) * termFw * dgg;
        const double qq = -2.0 * term2 * nut - dist2Inv * nut * nut * (kp.sa_cb1 * kar2Inv * (dfv2 - ft2 * dfv2 - fv2 * dft2 + dft2) - kp.sa_cw1 * dfw);
        *qqOut = fmax(qq, 0.0);
    }
    return (term1 + term2 * nut) * nut;
}

#endif
