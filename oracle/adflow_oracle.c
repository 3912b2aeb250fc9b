/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's Euler residual path, loop for loop in
 * the reference's own (face-scatter) order.  Each function cites the reference
 * lines it follows.  Pinned against the reference's own Fortran (oracle/_ref)
 * and the golden vectors in tests/golden/ by tests/test_oracle.py.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Arrays: Fortran order, bounds as the reference allocates them
 *   w (0:ib,0:jb,0:kb,nw)  p,gamma (0:ib,0:jb,0:kb)  sI (0:ie,1:je,1:ke,3)
 *   sJ (1:ie,0:je,1:ke,3)  sK (1:ie,1:je,0:ke,3)   porI (1:il,2:jl,2:kl) ...
 *   radI/J/K, dtl (1:ie,1:je,1:ke)   dw (0:ib,..,nw)   fw (0:ib,..,5)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int nx, ny, nz, nw;
    /* options / reference state (names of the reference's module variables) */
    double vis2, vis4, adis, acousticScaleFactor, rFil;
    double gammaInf, pInfCorr, rhoInf;
    int dirScaling, onlyRadii, iblankUsed;
    const double *w, *p, *gamma, *sI, *sJ, *sK;
    const int8_t *porI, *porJ, *porK;
    const int32_t* iblank;
    double *radI, *radJ, *radK, *dtl, *dw, *fw;
} oracle_block;

#define IL (b->nx + 1)
#define JL (b->ny + 1)
#define KL (b->nz + 1)
#define IE (b->nx + 2)
#define JE (b->ny + 2)
#define KE (b->nz + 2)
#define IB (b->nx + 3)
#define JB (b->ny + 3)
#define KB (b->nz + 3)
/* cell-box arrays (0:ib,0:jb,0:kb[,l]) */
#define CIDX(i, j, k) ((size_t)(i) + (size_t)(IB + 1) * ((size_t)(j) + (size_t)(JB + 1) * (size_t)(k)))
#define NBOX ((size_t)(IB + 1) * (JB + 1) * (KB + 1))
#define W(i, j, k, l) b->w[CIDX(i, j, k) + (size_t)(l) * NBOX]
#define P(i, j, k) b->p[CIDX(i, j, k)]
#define GAM(i, j, k) b->gamma[CIDX(i, j, k)]
#define DW(i, j, k, l) b->dw[CIDX(i, j, k) + (size_t)(l) * NBOX]
#define FW(i, j, k, l) b->fw[CIDX(i, j, k) + (size_t)(l) * NBOX]
/* face normals */
#define SI(i, j, k, d) b->sI[(size_t)(i) + (size_t)(IE + 1) * ((size_t)((j) - 1) + (size_t)JE * (size_t)((k) - 1)) + (size_t)(d) * (IE + 1) * JE * KE]
#define SJ(i, j, k, d) b->sJ[(size_t)((i) - 1) + (size_t)IE * ((size_t)(j) + (size_t)(JE + 1) * (size_t)((k) - 1)) + (size_t)(d) * IE * (JE + 1) * KE]
#define SK(i, j, k, d) b->sK[(size_t)((i) - 1) + (size_t)IE * ((size_t)((j) - 1) + (size_t)JE * (size_t)(k)) + (size_t)(d) * IE * JE * (KE + 1)]
#define PORI(i, j, k) b->porI[(size_t)((i) - 1) + (size_t)IL * ((size_t)((j) - 2) + (size_t)b->ny * (size_t)((k) - 2))]
#define PORJ(i, j, k) b->porJ[(size_t)((i) - 2) + (size_t)b->nx * ((size_t)((j) - 1) + (size_t)JL * (size_t)((k) - 2))]
#define PORK(i, j, k) b->porK[(size_t)((i) - 2) + (size_t)b->nx * ((size_t)((j) - 2) + (size_t)b->ny * (size_t)((k) - 1))]
#define HIDX(i, j, k) ((size_t)((i) - 1) + (size_t)IE * ((size_t)((j) - 1) + (size_t)JE * (size_t)((k) - 1)))
#define RADI(i, j, k) b->radI[HIDX(i, j, k)]
#define RADJ(i, j, k) b->radJ[HIDX(i, j, k)]
#define RADK(i, j, k) b->radK[HIDX(i, j, k)]
#define DTL(i, j, k) b->dtl[HIDX(i, j, k)]

enum { noFlux = -1, boundFlux = 0, normalFlux = 1 };
enum { irho = 0, ivx = 1, ivy = 2, ivz = 3, irhoE = 4 };

/* solverUtils::timeStep_block, src/solver/solverUtils.F90:43-356 (noPrecond, inviscid, steady) */
void oracle_time_step(oracle_block* b)
{
    const double plim = 0.001 * b->pInfCorr;
    const double clim2 = 0.000001 * b->gammaInf * b->pInfCorr / b->rhoInf;
    const double eps = 1.e-25;
    for (int k = 1; k <= KE; ++k)
        for (int j = 1; j <= JE; ++j)
            for (int i = 1; i <= IE; ++i) {                       /* :130-235 */
                const double uux = W(i, j, k, ivx), uuy = W(i, j, k, ivy), uuz = W(i, j, k, ivz);
                double cc2 = GAM(i, j, k) * P(i, j, k) / W(i, j, k, irho);
                cc2 = fmax(cc2, clim2);
                double sx = SI(i - 1, j, k, 0) + SI(i, j, k, 0), sy = SI(i - 1, j, k, 1) + SI(i, j, k, 1),
                       sz = SI(i - 1, j, k, 2) + SI(i, j, k, 2);
                double ri = 0.5 * (fabs(uux * sx + uuy * sy + uuz * sz) +
                                   b->acousticScaleFactor * sqrt(cc2 * (sx * sx + sy * sy + sz * sz)));
                sx = SJ(i, j - 1, k, 0) + SJ(i, j, k, 0); sy = SJ(i, j - 1, k, 1) + SJ(i, j, k, 1);
                sz = SJ(i, j - 1, k, 2) + SJ(i, j, k, 2);
                double rj = 0.5 * (fabs(uux * sx + uuy * sy + uuz * sz) +
                                   b->acousticScaleFactor * sqrt(cc2 * (sx * sx + sy * sy + sz * sz)));
                sx = SK(i, j, k - 1, 0) + SK(i, j, k, 0); sy = SK(i, j, k - 1, 1) + SK(i, j, k, 1);
                sz = SK(i, j, k - 1, 2) + SK(i, j, k, 2);
                double rk = 0.5 * (fabs(uux * sx + uuy * sy + uuz * sz) +
                                   b->acousticScaleFactor * sqrt(cc2 * (sx * sx + sy * sy + sz * sz)));
                if (!b->onlyRadii) DTL(i, j, k) = ri + rj + rk;
                if (b->dirScaling) {                               /* :178-199 */
                    ri = fmax(ri, eps); rj = fmax(rj, eps); rk = fmax(rk, eps);
                    const double rij = pow(ri / rj, b->adis), rjk = pow(rj / rk, b->adis), rki = pow(rk / ri, b->adis);
                    RADI(i, j, k) = ri * (1.0 + 1.0 / rij + rki);
                    RADJ(i, j, k) = rj * (1.0 + 1.0 / rjk + rij);
                    RADK(i, j, k) = rk * (1.0 + 1.0 / rki + rjk);
                } else {
                    RADI(i, j, k) = ri; RADJ(i, j, k) = rj; RADK(i, j, k) = rk;
                }
            }
    if (b->onlyRadii) return;
    for (int k = 2; k <= KL; ++k)                                  /* :336-352 */
        for (int j = 2; j <= JL; ++j)
            for (int i = 2; i <= IL; ++i) {
                const double dpi = fabs(P(i + 1, j, k) - 2.0 * P(i, j, k) + P(i - 1, j, k)) /
                                   (P(i + 1, j, k) + 2.0 * P(i, j, k) + P(i - 1, j, k) + plim);
                const double dpj = fabs(P(i, j + 1, k) - 2.0 * P(i, j, k) + P(i, j - 1, k)) /
                                   (P(i, j + 1, k) + 2.0 * P(i, j, k) + P(i, j - 1, k) + plim);
                const double dpk = fabs(P(i, j, k + 1) - 2.0 * P(i, j, k) + P(i, j, k - 1)) /
                                   (P(i, j, k + 1) + 2.0 * P(i, j, k) + P(i, j, k - 1) + plim);
                const double rfl = 1.0 / (1.0 + 2.0 * (dpi + dpj + dpk));
                DTL(i, j, k) = rfl / DTL(i, j, k);
            }
}

/* one face of fluxes::inviscidCentralFlux, src/solver/fluxes.F90:52-129 */
static void central_face(oracle_block* b, int i, int j, int k, int i2, int j2, int k2, double sx, double sy, double sz, int por)
{
    double vnp = W(i2, j2, k2, ivx) * sx + W(i2, j2, k2, ivy) * sy + W(i2, j2, k2, ivz) * sz;
    double vnm = W(i, j, k, ivx) * sx + W(i, j, k, ivy) * sy + W(i, j, k, ivz) * sz;
    double porVel = 1.0, porFlux = 0.5;
    if (por == noFlux) porFlux = 0.0;
    if (por == boundFlux) { porVel = 0.0; vnp = 0.0; vnm = 0.0; }
    porVel *= porFlux;
    const double qsp = vnp * porVel, qsm = vnm * porVel;
    const double rqsp = qsp * W(i2, j2, k2, irho), rqsm = qsm * W(i, j, k, irho);
    const double pa = porFlux * (P(i2, j2, k2) + P(i, j, k));
    double fs = rqsp + rqsm;
    DW(i2, j2, k2, irho) -= fs; DW(i, j, k, irho) += fs;
    fs = rqsp * W(i2, j2, k2, ivx) + rqsm * W(i, j, k, ivx) + pa * sx;
    DW(i2, j2, k2, ivx) -= fs; DW(i, j, k, ivx) += fs;
    fs = rqsp * W(i2, j2, k2, ivy) + rqsm * W(i, j, k, ivy) + pa * sy;
    DW(i2, j2, k2, ivy) -= fs; DW(i, j, k, ivy) += fs;
    fs = rqsp * W(i2, j2, k2, ivz) + rqsm * W(i, j, k, ivz) + pa * sz;
    DW(i2, j2, k2, ivz) -= fs; DW(i, j, k, ivz) += fs;
    fs = qsp * W(i2, j2, k2, irhoE) + qsm * W(i, j, k, irhoE) + porFlux * (vnp * P(i2, j2, k2) + vnm * P(i, j, k));
    DW(i2, j2, k2, irhoE) -= fs; DW(i, j, k, irhoE) += fs;
}

/* fluxes::inviscidCentralFlux, src/solver/fluxes.F90:4-401 (non-moving block) */
void oracle_central_flux(oracle_block* b)
{
    for (int k = 2; k <= KL; ++k)
        for (int j = 2; j <= JL; ++j)
            for (int i = 1; i <= IL; ++i)
                central_face(b, i, j, k, i + 1, j, k, SI(i, j, k, 0), SI(i, j, k, 1), SI(i, j, k, 2), PORI(i, j, k));
    for (int k = 2; k <= KL; ++k)
        for (int j = 1; j <= JL; ++j)
            for (int i = 2; i <= IL; ++i)
                central_face(b, i, j, k, i, j + 1, k, SJ(i, j, k, 0), SJ(i, j, k, 1), SJ(i, j, k, 2), PORJ(i, j, k));
    for (int k = 1; k <= KL; ++k)
        for (int j = 2; j <= JL; ++j)
            for (int i = 2; i <= IL; ++i)
                central_face(b, i, j, k, i, j, k + 1, SK(i, j, k, 0), SK(i, j, k, 1), SK(i, j, k, 2), PORK(i, j, k));
}

/* one face of fluxes::inviscidDissFluxScalar, src/solver/fluxes.F90:1204-1272; (di,dj,dk) = unit index step */
static void jst_face(oracle_block* b, int i, int j, int k, int di, int dj, int dk, double rrad, double dssL, double dssR,
                     double fis2, double fis4)
{
    const int i2 = i + di, j2 = j + dj, k2 = k + dk, i3 = i + 2 * di, j3 = j + 2 * dj, k3 = k + 2 * dk, i0 = i - di, j0 = j - dj,
              k0 = k - dk;
    const double dis2 = fis2 * rrad * fmin(0.25, fmax(dssL, dssR));
    const double t = fis4 * rrad - dis2;
    const double dis4 = t > 0.0 ? t : 0.0;                          /* myDim, utils.F90:470-480 */
    double ddw, fs;
    ddw = W(i2, j2, k2, irho) - W(i, j, k, irho);
    fs = dis2 * ddw - dis4 * (W(i3, j3, k3, irho) - W(i0, j0, k0, irho) - 3.0 * ddw);
    FW(i2, j2, k2, irho) += fs; FW(i, j, k, irho) -= fs;
    for (int l = ivx; l <= ivz; ++l) {
        ddw = W(i2, j2, k2, l) * W(i2, j2, k2, irho) - W(i, j, k, l) * W(i, j, k, irho);
        fs = dis2 * ddw - dis4 * (W(i3, j3, k3, l) * W(i3, j3, k3, irho) - W(i0, j0, k0, l) * W(i0, j0, k0, irho) - 3.0 * ddw);
        FW(i2, j2, k2, l) += fs; FW(i, j, k, l) -= fs;
    }
    ddw = (W(i2, j2, k2, irhoE) + P(i2, j2, k2)) - (W(i, j, k, irhoE) + P(i, j, k));
    fs = dis2 * ddw - dis4 * ((W(i3, j3, k3, irhoE) + P(i3, j3, k3)) - (W(i0, j0, k0, irhoE) + P(i0, j0, k0)) - 3.0 * ddw);
    FW(i2, j2, k2, irhoE) += fs; FW(i, j, k, irhoE) -= fs;
}

/* fluxes::inviscidDissFluxScalar, src/solver/fluxes.F90:1049-1436 (Euler: pressure sensor) */
void oracle_diss_scalar(oracle_block* b)
{
    if (fabs(b->rFil) < 1.e-10) return;                             /* :1085 */
    const double sslim = 0.001 * b->pInfCorr;                       /* :1099 */
    const size_t nh = (size_t)IE * JE * KE;
    double* dss = (double*)malloc(sizeof(double) * nh * 3);
    for (int k = 1; k <= KE; ++k)                                   /* :1140-1165 */
        for (int j = 1; j <= JE; ++j)
            for (int i = 1; i <= IE; ++i) {
                const double s0 = P(i, j, k);
                dss[HIDX(i, j, k)] = fabs((P(i + 1, j, k) - 2.0 * s0 + P(i - 1, j, k)) / (P(i + 1, j, k) + 2.0 * s0 + P(i - 1, j, k) + sslim));
                dss[HIDX(i, j, k) + nh] = fabs((P(i, j + 1, k) - 2.0 * s0 + P(i, j - 1, k)) / (P(i, j + 1, k) + 2.0 * s0 + P(i, j - 1, k) + sslim));
                dss[HIDX(i, j, k) + 2 * nh] = fabs((P(i, j, k + 1) - 2.0 * s0 + P(i, j, k - 1)) / (P(i, j, k + 1) + 2.0 * s0 + P(i, j, k - 1) + sslim));
            }
    const double fis2 = b->rFil * b->vis2, fis4 = b->rFil * b->vis4, sfil = 1.0 - b->rFil;
    for (size_t n = 0; n < NBOX * 5; ++n) b->fw[n] *= sfil;         /* :1193  fw = sfil*fw */
    for (int k = 2; k <= KL; ++k)                                   /* :1204-1272 */
        for (int j = 2; j <= JL; ++j)
            for (int i = 1; i <= IL; ++i) {
                const double ppor = (PORI(i, j, k) == normalFlux) ? 0.5 : 0.0;
                jst_face(b, i, j, k, 1, 0, 0, ppor * (RADI(i, j, k) + RADI(i + 1, j, k)), dss[HIDX(i, j, k)], dss[HIDX(i + 1, j, k)], fis2, fis4);
            }
    for (int k = 2; k <= KL; ++k)                                   /* :1277-1350 */
        for (int j = 1; j <= JL; ++j)
            for (int i = 2; i <= IL; ++i) {
                const double ppor = (PORJ(i, j, k) == normalFlux) ? 0.5 : 0.0;
                jst_face(b, i, j, k, 0, 1, 0, ppor * (RADJ(i, j, k) + RADJ(i, j + 1, k)), dss[HIDX(i, j, k) + nh], dss[HIDX(i, j + 1, k) + nh], fis2, fis4);
            }
    for (int k = 1; k <= KL; ++k)                                   /* :1355-1430 */
        for (int j = 2; j <= JL; ++j)
            for (int i = 2; i <= IL; ++i) {
                const double ppor = (PORK(i, j, k) == normalFlux) ? 0.5 : 0.0;
                jst_face(b, i, j, k, 0, 0, 1, ppor * (RADK(i, j, k) + RADK(i, j, k + 1)), dss[HIDX(i, j, k) + 2 * nh], dss[HIDX(i, j, k + 1) + 2 * nh], fis2, fis4);
            }
    free(dss);
}

/* blockette::blockResCore for Euler + scalar JST, src/NKSolver/blockette.F90:755-852:
 * timeStep_block, initres_block (steady ground level: dw = 0), fw = 0, central,
 * scalar dissipation, sumDwAndFw (adjointExtra.F90:638-659) */
void oracle_block_res_euler_scalar(oracle_block* b)
{
    oracle_time_step(b);
    for (int l = 0; l < 5; ++l)
        for (int k = 2; k <= KL; ++k)
            for (int j = 2; j <= JL; ++j)
                for (int i = 2; i <= IL; ++i) DW(i, j, k, l) = 0.0;
    memset(b->fw, 0, sizeof(double) * NBOX * 5);
    oracle_central_flux(b);
    oracle_diss_scalar(b);
    for (int l = 0; l < 5; ++l)
        for (int k = 2; k <= KL; ++k)
            for (int j = 2; j <= JL; ++j)
                for (int i = 2; i <= IL; ++i) {
                    const double bl = b->iblank ? (double)b->iblank[CIDX(i, j, k)] : 1.0;
                    DW(i, j, k, l) = (DW(i, j, k, l) + FW(i, j, k, l)) * (bl > 0.0 ? bl : 0.0);
                }
}
