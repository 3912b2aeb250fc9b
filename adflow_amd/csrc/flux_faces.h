// Face fluxes of the inviscid residual shared by the cell-gather kernel (kernels_inviscid.hip) and the k-marching
// kernel for matrix dissipation / Roe upwind (kernels_inviscid_march.hip).  A `Line` holds the primitive state of up
// to five consecutive cells along one index direction; the face functions act on the face between positions l and l+1.
//
// Reference semantics:
//   central flux        fluxes::inviscidCentralFlux     src/solver/fluxes.F90:4-401
//   scalar JST          fluxes::inviscidDissFluxScalar  src/solver/fluxes.F90:1049-1436
//   matrix JST          fluxes::inviscidDissFluxMatrix  src/solver/fluxes.F90:403-1047
//   Roe upwind (MUSCL)  fluxes::inviscidUpwindFlux      src/solver/fluxes.F90:1438-2532
#ifndef ADFLOW_FLUX_FACES_H
#define ADFLOW_FLUX_FACES_H
#include "internal.h"

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
// sFace: grid velocity through the face (moving blocks), 0 at rest
__device__ __forceinline__ void central_face(const Line& L, int l, double sx, double sy, double sz, int por,
                                             double sign, double acc[5], double sFace = 0.0)
{
    const int r = l + 1;
    double vnp = L.u[r] * sx + L.v[r] * sy + L.w[r] * sz;
    double vnm = L.u[l] * sx + L.v[l] * sy + L.w[l] * sz;
    double porVel = 1.0, porFlux = 0.5;
    if (por == ADF_POR_NOFLUX) porFlux = 0.0;
    if (por == ADF_POR_BOUND) {
        porVel = 0.0;
        vnp = sFace;
        vnm = sFace;
    }
    porVel *= porFlux;
    const double qsp = (vnp - sFace) * porVel, qsm = (vnm - sFace) * porVel;
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
// approx: inviscidDissFluxScalarApprox (fluxes.F90:3861-4342): dis2 + sigma*fis4*rrad on the first difference only
__device__ __forceinline__ void jst_scalar_face(const Line& L, int l, double rrad, double dssL, double dssR,
                                                double fis2, double fis4, double sign, double acc[5], bool approx = false,
                                                double sigma = 0.0)
{
    const int r = l + 1, ll = l - 1, rr = l + 2;
    double dis2 = fis2 * rrad * fmin(0.25, fmax(dssL, dssR));
    double dis4 = fmax(fis4 * rrad - dis2, 0.0);   // myDim, utils.F90:470-480
    if (approx) { dis2 = dis2 + sigma * fis4 * rrad; dis4 = 0.0; }
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

// matrix JST sensor (pressure, omega = 0.5 blending; fluxes.F90:495-508)
__device__ __forceinline__ double mat_sensor(double pm, double p0, double pp, double plim)
{
    return fabs((pp - 2.0 * p0 + pm) /
                (0.5 * (pp + 2.0 * p0 + pm) + 0.5 * (fabs(pp - p0) + fabs(p0 - pm)) + plim));
}

// shared tail of the matrix-dissipation and Roe fluxes: |A| applied to the
// conservative difference (dr,dru,drv,drw,dre)  (fluxes.F90:626-690, 2469-2501)
__device__ __forceinline__ void absA_times_dw(double lam1, double lam2, double lam3, double gm1, double alphaAvg,
                                              double uAvg, double vAvg, double wAvg, double hAvg, double unAvg,
                                              double ovaAvg, double ova2Avg, double sx, double sy, double sz,
                                              double dr, double dru, double drv, double drw, double dre, double f[5])
{
    const double abv1 = 0.5 * (lam1 + lam2);
    const double abv2 = 0.5 * (lam1 - lam2);
    const double abv3 = abv1 - lam3;
    const double abv4 = gm1 * (alphaAvg * dr - uAvg * dru - vAvg * drv - wAvg * drw + dre);   // - gm53*drk, drk = 0 (SA)
    const double abv5 = sx * dru + sy * drv + sz * drw - unAvg * dr;
    const double abv6 = abv3 * abv4 * ova2Avg + abv2 * abv5 * ovaAvg;
    const double abv7 = abv2 * abv4 * ovaAvg + abv3 * abv5;
    f[0] = lam3 * dr + abv6;
    f[1] = lam3 * dru + uAvg * abv6 + sx * abv7;
    f[2] = lam3 * drv + vAvg * abv6 + sy * abv7;
    f[3] = lam3 * drw + wAvg * abv6 + sz * abv7;
    f[4] = lam3 * dre + hAvg * abv6 + unAvg * abv7;
}

// matrix JST dissipative flux through face (l | l+1)  (fluxes.F90:523-690)
__device__ __forceinline__ void jst_matrix_face(const Line& L, const double gam[5], int l, double nx, double ny, double nz,
                                                int por, double dssL, double dssR, double fis2, double fis4, double sign,
                                                double acc[5], bool coarse = false, bool approx = false, double sigma = 0.0,
                                                double sFace = 0.0)
{
    const int r = l + 1, ll = l - 1, rr = l + 2;
    const double ppor = (por == ADF_POR_NORMAL) ? 1.0 : 0.0;
    // coarse multigrid levels (inviscidDissFluxMatrixCoarse, fluxes.F90:5205-5430): first
    // differences only, dis0 = rFil*vis2Coarse*ppor passed in fis2, no sensor
    double dis2 = coarse ? ppor * fis2 : ppor * fis2 * fmin(0.25, fmax(dssL, dssR));
    double dis4 = coarse ? 0.0 : fmax(ppor * fis4 - dis2, 0.0);
    if (approx) { dis2 = dis2 + sigma * fis4 * ppor; dis4 = 0.0; }   // inviscidDissFluxMatrixApprox (fluxes.F90:4462)
    double ddw;
    ddw = L.rho[r] - L.rho[l];
    const double dr = dis2 * ddw - dis4 * (L.rho[rr] - L.rho[ll] - 3.0 * ddw);
    ddw = L.rho[r] * L.u[r] - L.rho[l] * L.u[l];
    const double dru = dis2 * ddw - dis4 * (L.rho[rr] * L.u[rr] - L.rho[ll] * L.u[ll] - 3.0 * ddw);
    ddw = L.rho[r] * L.v[r] - L.rho[l] * L.v[l];
    const double drv = dis2 * ddw - dis4 * (L.rho[rr] * L.v[rr] - L.rho[ll] * L.v[ll] - 3.0 * ddw);
    ddw = L.rho[r] * L.w[r] - L.rho[l] * L.w[l];
    const double drw = dis2 * ddw - dis4 * (L.rho[rr] * L.w[rr] - L.rho[ll] * L.w[ll] - 3.0 * ddw);
    ddw = L.e[r] - L.e[l];
    const double dre = dis2 * ddw - dis4 * (L.e[rr] - L.e[ll] - 3.0 * ddw);

    const double gammaAvg = 0.5 * (gam[r] + gam[l]);
    const double gm1 = gammaAvg - 1.0;
    const double ovgm1 = fastdiv(1.0, gm1);
    const double uAvg = 0.5 * (L.u[r] + L.u[l]);
    const double vAvg = 0.5 * (L.v[r] + L.v[l]);
    const double wAvg = 0.5 * (L.w[r] + L.w[l]);
    const double a2Avg = 0.5 * (fastdiv(gam[r] * L.p[r], L.rho[r]) + fastdiv(gam[l] * L.p[l], L.rho[l]));
    const double area = sqrt(nx * nx + ny * ny + nz * nz);
    const double tmp = fastdiv(1.0, fmax(1.e-25, area));
    const double sx = nx * tmp, sy = ny * tmp, sz = nz * tmp;
    const double alphaAvg = 0.5 * (uAvg * uAvg + vAvg * vAvg + wAvg * wAvg);
    const double hAvg = alphaAvg + ovgm1 * a2Avg;
    const double aAvg = sqrt(a2Avg);
    const double unAvg = uAvg * sx + vAvg * sy + wAvg * sz;
    const double ovaAvg = fastdiv(1.0, aAvg), ova2Avg = fastdiv(1.0, a2Avg);
    const double sface = sFace * tmp;         // fluxes.F90:616
    double lam1 = fabs(unAvg - sface + aAvg), lam2 = fabs(unAvg - sface - aAvg), lam3 = fabs(unAvg - sface);
    const double rrad = lam3 + aAvg;
    lam1 = fmax(lam1, 0.25 * rrad) * area;    // epsAcoustic
    lam2 = fmax(lam2, 0.25 * rrad) * area;
    lam3 = fmax(lam3, 0.025 * rrad) * area;   // epsShear
    double f[5];
    absA_times_dw(lam1, lam2, lam3, gm1, alphaAvg, uAvg, vAvg, wAvg, hAvg, unAvg, ovaAvg, ova2Avg, sx, sy, sz, dr, dru,
                  drv, drw, dre, f);
#pragma unroll
    for (int m = 0; m < 5; ++m) acc[m] += sign * f[m];
}

// MUSCL left/right state corrections (fluxes.F90:2103-2294 leftRightState)
__device__ __forceinline__ void muscl(int lim, double omk, double opk, double factMinmod, double du1, double du2, double du3,
                                      double& left, double& right)
{
    if (lim == ADFLOW_LIM_NONE) {
        left = omk * du1 + opk * du2;
        right = -omk * du3 - opk * du2;
        return;
    }
    const double epsLim = 1.e-10;
    // clamped denominators of the four slope ratios r = a / b (fluxes.F90:2167-2190)
    const double d1 = copysign(fmax(fabs(du1), epsLim), du1), d2 = copysign(fmax(fabs(du2), epsLim), du2),
                 d3 = copysign(fmax(fabs(du3), epsLim), du3);
    double rl1, rl2, rr1, rr2;
    if (lim == ADFLOW_LIM_VANALBADA) {
        // r (r + 1) / (r^2 + 1) with r = max(0, a / b) is a (a + b) / (a^2 + b^2) for a b > 0 and 0 otherwise: one
        // division per ratio instead of two (the kernel is bound by FP64 divisions)
        auto phi = [](double a, double b) { return (a * b > 0.0) ? fastdiv(a * (a + b), a * a + b * b) : 0.0; };
        rl1 = phi(du2, d1);
        rl2 = phi(du1, d2);
        rr1 = phi(du3, d2);
        rr2 = phi(du2, d3);
    } else {   // minmod
        rl1 = fmin(1.0, factMinmod * fmax(0.0, fastdiv(du2, d1)));
        rl2 = fmin(1.0, factMinmod * fmax(0.0, fastdiv(du1, d2)));
        rr1 = fmin(1.0, factMinmod * fmax(0.0, fastdiv(du3, d2)));
        rr2 = fmin(1.0, factMinmod * fmax(0.0, fastdiv(du2, d3)));
    }
    left = omk * rl1 * du1 + opk * rl2 * du2;
    right = -opk * rr1 * du2 - omk * rr2 * du3;
}

// Roe dissipation flux through face (l | l+1) from MUSCL-reconstructed
// primitive states (rho,u,v,w,p); fw(left) += flux, fw(right) -= flux
// (fluxes.F90:1790-1889 + riemannFlux :2296-2532)
__device__ __forceinline__ void roe_face(const Line& L, const double gam[5], int l, double nx, double ny, double nz, int por,
                                         int lim, double kappaCoef, double rFil, double gammaConstant, double sign,
                                         double acc[5], double sFace = 0.0)
{
    const int r = l + 1, ll = l - 1, rr = l + 2;
    double left[5], right[5];
    if (lim == ADFLOW_LIM_FIRST_ORDER) {
        left[0] = L.rho[l]; left[1] = L.u[l]; left[2] = L.v[l]; left[3] = L.w[l]; left[4] = L.p[l];
        right[0] = L.rho[r]; right[1] = L.u[r]; right[2] = L.v[r]; right[3] = L.w[r]; right[4] = L.p[r];
    } else {
        const double omk = 0.25 * (1.0 - kappaCoef), opk = 0.25 * (1.0 + kappaCoef);
        const double factMinmod = (3.0 - kappaCoef) / fmax(1.e-10, 1.0 - kappaCoef);
        const double* q[5] = {L.rho, L.u, L.v, L.w, L.p};
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            const double du1 = q[m][l] - q[m][ll];
            const double du2 = q[m][r] - q[m][l];
            const double du3 = q[m][rr] - q[m][r];
            double dl, dr_;
            muscl(lim, omk, opk, factMinmod, du1, du2, du3, dl, dr_);
            left[m] = dl + q[m][l];
            right[m] = dr_ + q[m][r];
        }
    }
    double porFlux = 0.5 * rFil;
    if (por == ADF_POR_NOFLUX || por == ADF_POR_BOUND) porFlux = 0.0;
    const double gammaFace = 0.5 * (gam[l] + gam[r]);
    const double gm1 = gammaFace - 1.0;
    const double z1l = sqrt(left[0]), z1r = sqrt(right[0]);
    double tmp = fastdiv(1.0, z1l + z1r);
    const double ovgm1 = 1.0 / (gammaConstant - 1.0);   // flowUtils::etot/eint, cpConstant (uniform: scalar unit)
    const double Etl = left[0] * (fastdiv(ovgm1 * left[4], left[0]) + 0.5 * (left[1] * left[1] + left[2] * left[2] + left[3] * left[3]));
    const double Etr = right[0] * (fastdiv(ovgm1 * right[4], right[0]) + 0.5 * (right[1] * right[1] + right[2] * right[2] + right[3] * right[3]));
    const double dr = right[0] - left[0];
    const double dru = right[0] * right[1] - left[0] * left[1];
    const double drv = right[0] * right[2] - left[0] * left[2];
    const double drw = right[0] * right[3] - left[0] * left[3];
    const double drE = Etr - Etl;
    const double uAvg = tmp * (z1l * left[1] + z1r * right[1]);
    const double vAvg = tmp * (z1l * left[2] + z1r * right[2]);
    const double wAvg = tmp * (z1l * left[3] + z1r * right[3]);
    const double hAvg = tmp * (fastdiv(Etl + left[4], z1l) + fastdiv(Etr + right[4], z1r));
    const double area = sqrt(nx * nx + ny * ny + nz * nz);
    tmp = fastdiv(1.0, fmax(1.e-25, area));
    const double sx = nx * tmp, sy = ny * tmp, sz = nz * tmp;
    const double alphaAvg = 0.5 * (uAvg * uAvg + vAvg * vAvg + wAvg * wAvg);
    const double a2Avg = fabs(gm1 * (hAvg - alphaAvg));
    const double aAvg = sqrt(a2Avg);
    double unAvg = uAvg * sx + vAvg * sy + wAvg * sz;
    const double ovaAvg = fastdiv(1.0, aAvg), ova2Avg = fastdiv(1.0, a2Avg);
    const double rFace = sFace * tmp;          // fluxes.F90:2420
    if (por == ADF_POR_BOUND) unAvg = rFace;
    const double eta = 0.5 * (fabs((left[1] - right[1]) * sx + (left[2] - right[2]) * sy + (left[3] - right[3]) * sz) +
                              fabs(sqrt(fastdiv(gammaFace * left[4], left[0])) - sqrt(fastdiv(gammaFace * right[4], right[0]))));
    double lam1 = fabs(unAvg - rFace + aAvg), lam2 = fabs(unAvg - rFace - aAvg), lam3 = fabs(unAvg - rFace);
    tmp = 2.0 * eta;
    if (lam1 < tmp) lam1 = eta + fastdiv(0.25 * lam1 * lam1, eta);
    if (lam2 < tmp) lam2 = eta + fastdiv(0.25 * lam2 * lam2, eta);
    if (lam3 < tmp) lam3 = eta + fastdiv(0.25 * lam3 * lam3, eta);
    lam1 *= area; lam2 *= area; lam3 *= area;
    double f[5];
    absA_times_dw(lam1, lam2, lam3, gm1, alphaAvg, uAvg, vAvg, wAvg, hAvg, unAvg, ovaAvg, ova2Avg, sx, sy, sz, dr, dru,
                  drv, drw, drE, f);
#pragma unroll
    for (int m = 0; m < 5; ++m) acc[m] += sign * (-porFlux * f[m]);
}

#endif
