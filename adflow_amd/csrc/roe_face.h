// Face arithmetic of the Roe marching kernels (kernels_roe_march.hip: second-order MUSCL on the fine level; kernels_pc_march.hip: the
// first-order form of the preconditioner matrix, plain and on dual numbers): central flux + Roe dissipation flux through ONE face.
// Reference semantics: fluxes::inviscidCentralFlux src/solver/fluxes.F90:52-129, riemannFlux :2296-2532 (regrouped, see
// kernels_roe_march.hip).  One guard per build: kernels_ad.hip includes it a second time inside namespace adj with `double` standing
// for the dual number -- what is geometry or an option is typed adf_real8 and stays a plain double there.
#ifndef ADF_AD_BUILD
#ifndef ADFLOW_ROE_FACE_H
#define ADFLOW_ROE_FACE_H
#define ADF_ROE_FACE_BODY
#endif
#else
#ifndef ADFLOW_ROE_FACE_H_AD
#define ADFLOW_ROE_FACE_H_AD
#define ADF_ROE_FACE_BODY
#endif
#endif
#ifdef ADF_ROE_FACE_BODY
#undef ADF_ROE_FACE_BODY

struct RCell { double rho, u, v, w, p, e; };

struct RmK {              // uniform scalars of a launch
    adf_real8 omk, opk, factMinmod, gam, gm1, ovgm1, porDiss;
    bool doDiss;
};

// Fluxes through the face between cells b and c (normal S = (nx,ny,nz) pointing from b to c, porosity code por):
//   fc: central flux, dw(b) += fc, dw(c) -= fc                              (fluxes.F90:52-129)
//   fd: Roe dissipation flux = -porFlux |A| (W_R - W_L), fw(b) += fd, fw(c) -= fd   (riemannFlux, fluxes.F90:2296-2501)
// L / R: reconstructed primitive states (rho, u, v, w, p) on the two sides of the face.
__device__ __forceinline__ void rm_face(const RmK& K, const RCell& b, const RCell& c, const double L[5], const double R[5], adf_real8 nx,
                                        adf_real8 ny, adf_real8 nz, int por, double fc[5], double fd[5])
{
    // ---- central
    {
        double vnp = c.u * nx + c.v * ny + c.w * nz;
        double vnm = b.u * nx + b.v * ny + b.w * nz;
        adf_real8 porVel = 1.0, porFlux = 0.5;
        if (por == ADF_POR_NOFLUX) porFlux = 0.0;
        if (por == ADF_POR_BOUND) { porVel = 0.0; vnp = 0.0; vnm = 0.0; }
        porVel *= porFlux;
        const double qsp = vnp * porVel, qsm = vnm * porVel;
        const double rqsp = qsp * c.rho, rqsm = qsm * b.rho;
        const double pa = porFlux * (c.p + b.p);
        fc[0] = rqsp + rqsm;
        fc[1] = rqsp * c.u + rqsm * b.u + pa * nx;
        fc[2] = rqsp * c.v + rqsm * b.v + pa * ny;
        fc[3] = rqsp * c.w + rqsm * b.w + pa * nz;
        fc[4] = qsp * c.e + qsm * b.e + porFlux * (vnp * c.p + vnm * b.p);
    }
    if (!K.doDiss) {
#pragma unroll
        for (int m = 0; m < 5; ++m) fd[m] = 0.0;
        return;
    }
    // ---- Roe
    adf_real8 porFlux = K.porDiss;                                // 0.5 rFil
    if (por == ADF_POR_NOFLUX || por == ADF_POR_BOUND) porFlux = 0.0;
    const double rsl = rsq_nr(L[0]), rsr = rsq_nr(R[0]);       // 1 / z1l, 1 / z1r
    const double z1l = L[0] * rsl, z1r = R[0] * rsr;
    const double zs = z1l + z1r;
    const double kl = 0.5 * (L[1] * L[1] + L[2] * L[2] + L[3] * L[3]), kr = 0.5 * (R[1] * R[1] + R[2] * R[2] + R[3] * R[3]);
    const double Etl = K.ovgm1 * L[4] + L[0] * kl, Etr = K.ovgm1 * R[4] + R[0] * kr;   // etot, cpConstant (flowUtils.F90:551-640)
    const double dr = R[0] - L[0];
    const double dru = R[0] * R[1] - L[0] * L[1];
    const double drv = R[0] * R[2] - L[0] * L[2];
    const double drw = R[0] * R[3] - L[0] * L[3];
    const double drE = Etr - Etl;
    const adf_real8 a2n = nx * nx + ny * ny + nz * nz;
    const adf_real8 ra = rsq_nr(fmax(a2n, 1.e-50));               // 1 / max(1e-25, area)
    const adf_real8 area = a2n * ra;
    const adf_real8 sx = nx * ra, sy = ny * ra, sz = nz * ra;
    // the sound speeds of the two states for the entropy fix: sqrt(gamma p / rho) = sqrt(gamma p) / sqrt(rho); its eta shares one
    // reciprocal with the Roe weights (1 / x = y / (x y))
    const double gpl = K.gam * L[4], gpr = K.gam * R[4];
    const double cl = gpl * rsq_nr(gpl) * rsl, cr = gpr * rsq_nr(gpr) * rsr;
    const double eta = 0.5 * (fabs((L[1] - R[1]) * sx + (L[2] - R[2]) * sy + (L[3] - R[3]) * sz) + fabs(cl - cr));
    const double etaC = fmax(eta, 1.e-290);                    // q4eta is only used where lam < 2 eta, i.e. eta > 0
    const double rze = rcp_nr(zs * etaC);
    const double tmp = rze * etaC;                             // 1 / (z1l + z1r)
    const double q4eta = 0.25 * (rze * zs);                    // 1 / (4 eta)
    const double wl = z1l * tmp, wr = z1r * tmp;
    const double uAvg = wl * L[1] + wr * R[1];
    const double vAvg = wl * L[2] + wr * R[2];
    const double wAvg = wl * L[3] + wr * R[3];
    const double hAvg = tmp * ((Etl + L[4]) * rsl + (Etr + R[4]) * rsr);
    const double alphaAvg = 0.5 * (uAvg * uAvg + vAvg * vAvg + wAvg * wAvg);
    const double a2Avg = fabs(K.gm1 * (hAvg - alphaAvg));
    const double ovaAvg = rsq_nr(a2Avg), ova2Avg = ovaAvg * ovaAvg, aAvg = a2Avg * ovaAvg;
    double unAvg = uAvg * sx + vAvg * sy + wAvg * sz;
    if (por == ADF_POR_BOUND) unAvg = 0.0;                     // rFace = 0: blocks at rest (moving blocks use the gather kernel)
    double lam1 = fabs(unAvg + aAvg), lam2 = fabs(unAvg - aAvg), lam3 = fabs(unAvg);
    const double two_eta = 2.0 * eta;
    // lam < 2 eta: lam <- eta + lam^2 / (4 eta).  Without selects: with lc = min(lam, 2 eta) the parabola is >= lam below 2 eta and
    // equals 2 eta <= lam from there on
    {
        const double l1 = fmin(lam1, two_eta), l2 = fmin(lam2, two_eta), l3 = fmin(lam3, two_eta);
        lam1 = fmax(lam1, eta + (l1 * l1) * q4eta);
        lam2 = fmax(lam2, eta + (l2 * l2) * q4eta);
        lam3 = fmax(lam3, eta + (l3 * l3) * q4eta);
    }
    lam1 *= area; lam2 *= area; lam3 *= area;
    const double abv1 = 0.5 * (lam1 + lam2);
    const double abv2 = 0.5 * (lam1 - lam2);
    const double abv3 = abv1 - lam3;
    const double abv4 = K.gm1 * (alphaAvg * dr - uAvg * dru - vAvg * drv - wAvg * drw + drE);
    const double abv5 = sx * dru + sy * drv + sz * drw - unAvg * dr;
    const double abv6 = abv3 * abv4 * ova2Avg + abv2 * abv5 * ovaAvg;
    const double abv7 = abv2 * abv4 * ovaAvg + abv3 * abv5;
    fd[0] = -porFlux * (lam3 * dr + abv6);
    fd[1] = -porFlux * (lam3 * dru + uAvg * abv6 + sx * abv7);
    fd[2] = -porFlux * (lam3 * drv + vAvg * abv6 + sy * abv7);
    fd[3] = -porFlux * (lam3 * drw + wAvg * abv6 + sz * abv7);
    fd[4] = -porFlux * (lam3 * drE + hAvg * abv6 + unAvg * abv7);
}

#endif
