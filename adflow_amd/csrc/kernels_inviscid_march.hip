// k-marching inviscid residual for MATRIX dissipation and ROE UPWIND (Euler, laminar, RANS; fine and coarse levels) and
// for scalar JST with the entropy sensor (laminar / RANS, fine level; Euler + scalar JST has its own pipelined kernel).
//
// The cell-gather kernel (kernels_inviscid.hip) evaluates the six faces of every cell, i.e. every face twice.  These
// two schemes are bound by FP64 arithmetic (eigenvalue scaling, MUSCL limiter: ~300 divisions per cell in the gather
// form), so the gather form pays twice.  Here the march of the Euler / scalar-JST kernel (kernels_euler_march.hip) is
// reused:
//   * a workgroup is 64 lanes (i) x 4 rows (j); every thread marches along k with a 4-cell register window, so a
//     k-face is evaluated once and closes the cell below / opens the cell above;
//   * a 64-lane wavefront covers the columns i0-2 .. i0+61 and produces 60 of them: the lane evaluates the face
//     (i-1 | i) from its neighbours' state (DPP lane shifts) and hands the flux of (i | i+1) over from lane+1, so an
//     i-face is evaluated once;
//   * both j-faces of a cell are evaluated by the cell (rows are different waves).
// Four face evaluations per cell instead of six, and ~45 instead of ~100 loads per cell.  The face functions are the
// ones of the gather kernel (flux_faces.h): same arithmetic, same order inside a face.
//
// gamma: the reference reads gamma(i,j,k); with the calorically perfect gas (cpConstant, the only cp model of the
// path) that array holds gammaConstant everywhere, which is what the faces use here.
//
// Reference semantics: fluxes.F90:4-401 (central), :403-1047 (matrix), :1438-2532 (upwind), :5205-5430 (matrix,
// coarse levels), residuals.F90:334-344 (final sum).  Roofline: FP64 VALU for upwind, HBM for matrix.
//
// The file compiles a second time for dual numbers (kernels_ad.hip, namespace adj: the exact linearisation of the adjoint with the
// scalar / matrix dissipation): state, sensor, spectral radii and residual are `double` (dual
// there), the face normals adf_real8 (plain in both builds).
#include "flux_faces.h"

#define IM_OUT 60          // must match EM_OUT / EM_BY of kernels_euler_march.hip: the tile table is shared
#define IM_BY 4
#ifdef ADF_AD_BUILD
#define IM_MINWG 1         // dual numbers: twice the registers
#else
#define IM_MINWG 2
#endif

struct MCell { double rho, u, v, w, e, p, s; };     // s: sensor variable of the scalar JST scheme (entropy for NS / RANS)

struct ImPtrs {
    GPTR(const double) w0; GPTR(const double) w1; GPTR(const double) w2; GPTR(const double) w3; GPTR(const double) w4;
    GPTR(const double) p;
    GPTR(const double) ss;      // scalar JST only
};

template <bool SCAL>
__device__ __forceinline__ MCell im_ld(const ImPtrs& m, unsigned o)
{
    MCell q;
    q.rho = ldg(m.w0, o); q.u = ldg(m.w1, o); q.v = ldg(m.w2, o); q.w = ldg(m.w3, o); q.e = ldg(m.w4, o);
    q.p = ldg(m.p, o);
    q.s = SCAL ? ldg(m.ss, o) : 0.0;
    return q;
}

__device__ __forceinline__ MCell im_up1(const MCell& q)
{
    MCell r;
    r.rho = lane_up1(q.rho); r.u = lane_up1(q.u); r.v = lane_up1(q.v); r.w = lane_up1(q.w); r.e = lane_up1(q.e); r.p = lane_up1(q.p);
    r.s = lane_up1(q.s);
    return r;
}

__device__ __forceinline__ MCell im_dn1(const MCell& q)
{
    MCell r;
    r.rho = lane_dn1(q.rho); r.u = lane_dn1(q.u); r.v = lane_dn1(q.v); r.w = lane_dn1(q.w); r.e = lane_dn1(q.e); r.p = lane_dn1(q.p);
    r.s = lane_dn1(q.s);
    return r;
}

// the four cells around a face as positions 0..3 of a Line; the face is between positions 1 and 2
__device__ __forceinline__ void im_line(const MCell& a, const MCell& b, const MCell& c, const MCell& d, Line& L)
{
    const MCell* q[4] = {&a, &b, &c, &d};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        L.rho[m] = q[m]->rho; L.u[m] = q[m]->u; L.v[m] = q[m]->v; L.w[m] = q[m]->w; L.e[m] = q[m]->e; L.p[m] = q[m]->p;
    }
    L.rho[4] = L.u[4] = L.v[4] = L.w[4] = L.e[4] = L.p[4] = 0.0;
}

struct ImFace {          // scalars of the dissipation shared by all faces of a launch
    double fis2, fis4, plim, gam;
    int lim, coarse, doDiss;
    double kappaCoef, rFil, gammaConstant;
    double sigma;        // APX: weight of the fourth-difference coefficient lumped onto the first difference
};

// fluxes through the face between b and c (stencil a b | c d) with normal (sx,sy,sz): central fc (dw(b) += fc,
// dw(c) -= fc) and dissipation fd in the convention fw(c) += fd, fw(b) -= fd.  dssB / dssC: matrix sensors of b and c.
// APX: the lumped dissipation of the preconditioner matrix (inviscidDissFluxScalarApprox / MatrixApprox, fluxes.F90:3861-4967)
template <int SCHEME, bool APX = false>
__device__ __forceinline__ void im_face(const ImFace& F, const MCell& a, const MCell& b, const MCell& c, const MCell& d, double sx,
                                        double sy, double sz, int por, double dssB, double dssC, double fc[5], double fd[5],
                                        double radSum = 0.0)
{
    Line L;
    im_line(a, b, c, d, L);
#pragma unroll
    for (int m = 0; m < 5; ++m) { fc[m] = 0.0; fd[m] = 0.0; }
    central_face(L, 1, sx, sy, sz, por, +1.0, fc);
    if (!F.doDiss) return;
    const double gam[5] = {F.gam, F.gam, F.gam, F.gam, F.gam};
    if (SCHEME == ADFLOW_DISS_SCALAR) {
        // scalar JST on the fine level: radSum = spectral radii of b and c in the direction of the face (fluxes.F90:1204-1272)
        const double rrad = (por == ADF_POR_NORMAL ? 0.5 : 0.0) * radSum;
        jst_scalar_face(L, 1, rrad, dssB, dssC, F.fis2, F.fis4, +1.0, fd, APX, F.sigma);
    } else if (SCHEME == ADFLOW_DISS_MATRIX) {
        if (F.coarse) jst_matrix_face(L, gam, 1, sx, sy, sz, por, 0.0, 0.0, F.fis2, 0.0, +1.0, fd, true);
        else jst_matrix_face(L, gam, 1, sx, sy, sz, por, dssB, dssC, F.fis2, F.fis4, +1.0, fd, false, APX, F.sigma);
    } else {
        // roe_face: fw(left) += flux, fw(right) -= flux with flux = -porFlux |A| dW; sign -1 gives the (right += ) form
        roe_face(L, gam, 1, sx, sy, sz, por, F.lim, F.kappaCoef, F.rFil, F.gammaConstant, -1.0, fd);
    }
}

// FW: persistent dissipation residual of the Runge-Kutta scheme; FINAL: dw = (dw + fw) iblank written here, otherwise dw and
// fw are left for the viscous kernel to complete (residual_block, residuals.F90:334-344)
// ADDV (without FW and FINAL): the viscous march ran first and left its flux sums in dw(2:5); they are added here, before iblank
// APX (scalar / matrix, fine level, without FW): the approximate residual of the preconditioner matrix -- the sensor of both schemes
// from the FROZEN values of referenceShockSensor in b.ss (entropy or pressure for the scalar scheme, pressure for the matrix scheme),
// the fourth-difference coefficient lumped onto the first difference
template <int SCHEME, bool FW, bool FINAL, bool ADDV = false, bool APX = false>
__global__ __launch_bounds__(64 * IM_BY, IM_MINWG) void k_inviscid_march(const BlkView* __restrict__ tab, const int4* __restrict__ tiles,
                                                                  KParams kp, int kch)
{
    const int4 t = tiles[blockIdx.x];
    if (t.x < 0) return;
    const BlkView& b = tab[t.x];
    const int lane = threadIdx.x;
    const int i = t.y * IM_OUT + lane;          // columns i0-2 .. i0+61
    const int j = 2 + t.z * IM_BY + (int)threadIdx.y;
    const int k0 = 2 + t.w * kch;
    const int k1 = (k0 + kch - 1 < b.kl) ? k0 + kch - 1 : b.kl;
    const bool out = (lane >= 2 && lane <= 61 && i <= b.il && j <= b.jl);
    const int ic = (i < b.ib) ? i : b.ib, jc = (j < b.jl) ? j : b.jl;
    const long nb = b.nbox;
    const unsigned sj = 8u * (unsigned)b.ldi, sk = 8u * (unsigned)b.ldk;
    unsigned c = 8u * (unsigned)(ic + jc * b.ldi + k0 * b.ldk);

    ImPtrs m;
    m.w0 = (GPTR(const double))b.w; m.w1 = m.w0 + nb; m.w2 = m.w1 + nb; m.w3 = m.w2 + nb; m.w4 = m.w3 + nb;
    m.p = (GPTR(const double))b.p;
    m.ss = (GPTR(const double))b.ss;
    constexpr bool SCAL = (SCHEME == ADFLOW_DISS_SCALAR);       // NS / RANS, fine level: entropy sensor from b.ss, radii from the time step
    constexpr bool LDS_ = SCAL || APX;                           // the cells carry a sensor value from b.ss
    GPTR(const double) radI = (GPTR(const double))b.radI;
    GPTR(const double) radJ = (GPTR(const double))b.radJ;
    GPTR(const double) radK = (GPTR(const double))b.radK;
    GPTR(const adf_real8) sIx = (GPTR(const adf_real8))b.sI; GPTR(const adf_real8) sIy = sIx + nb; GPTR(const adf_real8) sIz = sIy + nb;
    GPTR(const adf_real8) sJx = (GPTR(const adf_real8))b.sJ; GPTR(const adf_real8) sJy = sJx + nb; GPTR(const adf_real8) sJz = sJy + nb;
    GPTR(const adf_real8) sKx = (GPTR(const adf_real8))b.sK; GPTR(const adf_real8) sKy = sKx + nb; GPTR(const adf_real8) sKz = sKy + nb;
    GPTR(const uint8_t) flags = (GPTR(const uint8_t))b.flags;
    GPTR(double) dw = (GPTR(double))b.dw;
    GPTR(double) fw = (GPTR(double))b.fw;
    GPTR(const double) wr = (GPTR(const double))b.wr;

    ImFace F;
    F.doDiss = fabs(kp.rFil) >= 1.e-10;
    F.coarse = !kp.fineGrid;
    F.fis2 = F.coarse ? kp.rFil * kp.vis2Coarse : kp.rFil * kp.vis2;     // coarse: dis0 of inviscidDissFluxMatrixCoarse
    F.fis4 = kp.rFil * kp.vis4;
    // sslim of the entropy sensor (scalar JST of NS / RANS) / plim of the pressure sensor (matrix; scalar JST of Euler, APX only)
    F.plim = (SCAL && kp.viscous) ? 0.001 * kp.pInfCorr / pow(kp.rhoInf, kp.gammaInf) : 0.001 * kp.pInfCorr;
    F.sigma = kp.sigma;
    F.gam = kp.gammaConstant;
    F.lim = (kp.fineGrid && !kp.lumpedDiss) ? kp.limiter : ADFLOW_LIM_FIRST_ORDER;           // fluxes.F90:1531-1538
    F.kappaCoef = kp.kappaCoef; F.rFil = kp.rFil; F.gammaConstant = kp.gammaConstant;
    const bool sens = (SCHEME != ADFLOW_UPWIND) && F.doDiss && !F.coarse;
    auto sensor = [&](const MCell& a, const MCell& q, const MCell& d) {
        if (SCAL) return jst_sensor(a.s, q.s, d.s, F.plim);
        return APX ? mat_sensor(a.s, q.s, d.s, F.plim) : mat_sensor(a.p, q.p, d.p, F.plim);
    };

    // window k-2 .. k+1 of the own column
    MCell qm2 = im_ld<LDS_>(m, c - 2 * sk), qm1 = im_ld<LDS_>(m, c - sk), q0 = im_ld<LDS_>(m, c);
    int flagm = flags[(c - sk) >> 3];
    double dssKm = sens ? sensor(qm2, qm1, q0) : 0.0;
    double radKm = SCAL ? ldg(radK, c - sk) : 0.0;
    double accC[5] = {0, 0, 0, 0, 0}, accD[5] = {0, 0, 0, 0, 0};

    for (int k = k0; k <= k1 + 1; ++k) {
        const MCell qp1 = im_ld<LDS_>(m, c + sk);
        const int flag0 = flags[c >> 3];
        const double dssK0 = sens ? sensor(qm1, q0, qp1) : 0.0;
        const double radK0 = SCAL ? ldg(radK, c) : 0.0;
        double vsum[4] = {0, 0, 0, 0};       // ADDV: requested ahead of the face evaluation that precedes their use
        if (ADDV && k > k0 && out) {
#pragma unroll
            for (int l = 0; l < 4; ++l) vsum[l] = ldg((GPTR(const double))dw + (l + 1) * nb, c - sk);
        }
        // ---- everything else the step reads, requested HERE: the wave shares its SIMD with one other, and every request that stands
        //      in front of its own use is a latency nobody covers (round 4; the k and the i face below run on what has arrived)
        const adf_real8 nK[3] = {ldg(sKx, c - sk), ldg(sKy, c - sk), ldg(sKz, c - sk)};
        const adf_real8 nI[3] = {ldg(sIx, c - 8u), ldg(sIy, c - 8u), ldg(sIz, c - 8u)};
        const double radI0 = SCAL ? ldg(radI, c) : 0.0;
        const MCell qa = im_ld<LDS_>(m, c - 2 * sj), qb = im_ld<LDS_>(m, c - sj), qc = im_ld<LDS_>(m, c + sj), qd = im_ld<LDS_>(m, c + 2 * sj);
        const int flagJm = flags[(c - sj) >> 3];
        const double radJ0 = SCAL ? ldg(radJ, c) : 0.0, radJm = SCAL ? ldg(radJ, c - sj) : 0.0, radJp = SCAL ? ldg(radJ, c + sj) : 0.0;
        const adf_real8 nJm[3] = {ldg(sJx, c - sj), ldg(sJy, c - sj), ldg(sJz, c - sj)};
        const adf_real8 nJ[3] = {ldg(sJx, c), ldg(sJy, c), ldg(sJz, c)};
        __builtin_amdgcn_sched_barrier(0);
        // ---- k-face between cells k-1 and k (normal and porosity stored at cell k-1)
        double fc[5], fd[5];
        im_face<SCHEME, APX>(F, qm2, qm1, q0, qp1, nK[0], nK[1], nK[2], flg_porK((uint8_t)flagm), dssKm, dssK0, fc, fd, radKm + radK0);
        // ---- finish cell k-1 and write it
        if (k > k0 && out) {
            const unsigned cw = c - sk;
            const double blank = flg_blank((uint8_t)flagm);
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                double fwn = accD[l] - fd[l];
                if (FW) {
                    const double old = ldg(fw + l * nb, cw);
                    fwn = F.doDiss ? (kp.sfil * old + fwn) : old;
                }
                double d = accC[l] + fc[l];
                if (kp.coarseInit) d += ldg(wr + l * nb, cw);
                if (FINAL) {
                    if (FW && F.doDiss) stg(fw + l * nb, cw, fwn);
                    stg(dw + l * nb, cw, (d + fwn) * blank);
                } else if (FW) {
                    stg(fw + l * nb, cw, fwn);
                    stg(dw + l * nb, cw, d);
                } else if (ADDV) {
                    stg(dw + l * nb, cw, ((d + fwn) + (l > 0 ? vsum[l - 1] : 0.0)) * blank);
                } else {
                    stg(dw + l * nb, cw, (d + fwn) * blank);      // fw not persistent: the viscous kernel adds its part to dw(2:5) and re-applies iblank
                }
            }
        }
        if (k > k1) break;
        // ---- start cell k
#pragma unroll
        for (int l = 0; l < 5; ++l) { accC[l] = -fc[l]; accD[l] = fd[l]; }

        // ---- i-direction: this lane evaluates the face (i-1 | i); the face (i | i+1) comes from lane+1
        {
            const MCell qL = im_up1(q0), qLL = im_up1(qL), qR = im_dn1(q0);
            const int por = flg_porI((uint8_t)lane_up1(flag0));
            double d0 = 0.0, dL = 0.0;
            if (sens) {
                d0 = sensor(qL, q0, qR);
                dL = lane_up1(d0);
            }
            double radSum = 0.0;
            if (SCAL) radSum = lane_up1(radI0) + radI0;
            double gc[5], gd[5];
            im_face<SCHEME, APX>(F, qLL, qL, q0, qR, nI[0], nI[1], nI[2], por, dL, d0, gc, gd, radSum);
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                accC[l] += lane_dn1(gc[l]) - gc[l];     // + plus face, - minus face
                accD[l] += gd[l] - lane_dn1(gd[l]);     // minus face adds, plus face subtracts
            }
        }
        // ---- j-direction: both faces of the cell
        {
            const int porM = flg_porJ((uint8_t)flagJm), porP = flg_porJ((uint8_t)flag0);
            double dm = 0.0, d0 = 0.0, dp = 0.0;
            if (sens) {
                dm = sensor(qa, qb, q0);
                d0 = sensor(qb, q0, qc);
                dp = sensor(q0, qc, qd);
            }
            double rM = 0.0, rP = 0.0;
            if (SCAL) {
                rM = radJm + radJ0;
                rP = radJ0 + radJp;
            }
            double hc[5], hd[5];
            im_face<SCHEME, APX>(F, qa, qb, q0, qc, nJm[0], nJm[1], nJm[2], porM, dm, d0, hc, hd, rM);
#pragma unroll
            for (int l = 0; l < 5; ++l) { accC[l] -= hc[l]; accD[l] += hd[l]; }
            im_face<SCHEME, APX>(F, qb, q0, qc, qd, nJ[0], nJ[1], nJ[2], porP, d0, dp, hc, hd, rP);
#pragma unroll
            for (int l = 0; l < 5; ++l) { accC[l] += hc[l]; accD[l] -= hd[l]; }
        }
        // ---- advance the window
        qm2 = qm1; qm1 = q0; q0 = qp1;
        radKm = radK0;
        dssKm = dssK0;
        flagm = flag0;
        c += sk;
    }
}

#ifndef ADF_AD_BUILD
int g_inviscid_march = 2;      // tuning "inviscid_march": 0 = cell-gather kernel for matrix / upwind too, 1 = gather kernel for NS / RANS scalar JST only, 2 = marching form there as well (3.07 vs 3.33 ms per scalar-JST RANS evaluation, profiles/r02_as_ab_config3.txt)
#endif

template <int SCHEME>
static void launch_im(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, int kch, hipStream_t s)
{
    const dim3 blk(64, IM_BY, 1), grd(ntiles);
    const bool doDiss = fabs(kp.rFil) >= 1.e-10;
    const bool final_ = !(kp.viscous && doDiss);       // as launch_scheme of the gather kernel
#ifndef ADF_AD_BUILD
    if (kp.fwMode) {
        if (final_) hipLaunchKernelGGL((k_inviscid_march<SCHEME, true, true>), grd, blk, 0, s, tab, tiles, kp, kch);
        else hipLaunchKernelGGL((k_inviscid_march<SCHEME, true, false>), grd, blk, 0, s, tab, tiles, kp, kch);
        return;
    }
#endif
    // (the forward-mode passes have no persistent fw: block_res_state_d evaluates the whole residual)
    if (kp.dissApprox && SCHEME != ADFLOW_UPWIND) {
        // (the approximate upwind residual differs through its limiter only -- F.lim below -- and takes the kernels of the exact one)
        constexpr int S2 = (SCHEME == ADFLOW_UPWIND) ? ADFLOW_DISS_SCALAR : SCHEME;
        if (kp.viscFirst) hipLaunchKernelGGL((k_inviscid_march<S2, false, false, true, true>), grd, blk, 0, s, tab, tiles, kp, kch);
        else if (final_) hipLaunchKernelGGL((k_inviscid_march<S2, false, true, false, true>), grd, blk, 0, s, tab, tiles, kp, kch);
        else hipLaunchKernelGGL((k_inviscid_march<S2, false, false, false, true>), grd, blk, 0, s, tab, tiles, kp, kch);
        return;
    }
    if (kp.viscFirst) {
        hipLaunchKernelGGL((k_inviscid_march<SCHEME, false, false, true>), grd, blk, 0, s, tab, tiles, kp, kch);
    } else {
        if (final_) hipLaunchKernelGGL((k_inviscid_march<SCHEME, false, true>), grd, blk, 0, s, tab, tiles, kp, kch);
        else hipLaunchKernelGGL((k_inviscid_march<SCHEME, false, false>), grd, blk, 0, s, tab, tiles, kp, kch);
    }
}

#ifndef ADF_AD_BUILD
extern int g_march_kch;
#endif

// matrix dissipation / Roe upwind over the tile table of the level (the table of the Euler marching kernel)
void launch_inviscid_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s)
{
    if (ntiles <= 0) return;
    if (kp.spaceDiscr == ADFLOW_DISS_MATRIX) launch_im<ADFLOW_DISS_MATRIX>(tab, tiles, ntiles, kp, ::g_march_kch, s);
#ifndef ADF_AD_BUILD
    else if (kp.spaceDiscr == ADFLOW_UPWIND) launch_im<ADFLOW_UPWIND>(tab, tiles, ntiles, kp, ::g_march_kch, s);
#else
    else if (kp.spaceDiscr == ADFLOW_UPWIND) return;      // (forward mode: the upwind scheme is k_roe_march's or the gather kernel's)
#endif
    else launch_im<ADFLOW_DISS_SCALAR>(tab, tiles, ntiles, kp, ::g_march_kch, s);     // NS / RANS on the fine level only (caller)
}

#ifndef ADF_AD_BUILD
// the shared tile table has IM_BY rows per tile unless the Euler kernel was switched to 8 rows (tuning march_by)
int inviscid_march_enabled() { return g_inviscid_march; }
#endif
