// k-marching inviscid residual for the second-order ROE UPWIND scheme on the fine level: central flux + MUSCL
// reconstruction (no limiter / van Albada / minmod) + Roe flux with the 1-D entropy fix, every face evaluated ONCE in i
// and k, the reconstruction evaluated once per CELL and direction.
//
// Reference semantics: fluxes::inviscidCentralFlux  src/solver/fluxes.F90:4-401
//                      fluxes::inviscidUpwindFlux   src/solver/fluxes.F90:1438-2532 (leftRightState :2103-2294,
//                      riemannFlux :2296-2532), final sum residuals.F90:334-344.
//
// Why a second kernel next to k_inviscid_march<upwind>: that one evaluates leftRightState per FACE as the reference does
// (20 limiter divisions per face, four faces per cell) and measures ~4400 FP64 issue slots per cell -- it is bound by
// FP64 issue, not by HBM.  Here the algebra is regrouped (results agree to rounding, not bitwise):
//   * leftRightState of face (c | c+1) gives  left  = w_c     + omk A_c     + opk B_c
//                                             right = w_{c+1} - opk A_{c+1} - omk B_{c+1}
//     with A_c = f(d+ / d-) d-,  B_c = f(d- / d+) d+  and d-, d+ the two differences of cell c in that direction: both
//     states of a cell are functions of (w_{c-1}, w_c, w_{c+1}) only.  A thread therefore reconstructs ITS cell once per
//     direction (two limiter values per variable instead of four per face and variable) and the face takes the left
//     state from the cell below (k: carried in registers; i: DPP lane shift; j: exchanged through LDS);
//   * the two limiter denominators of a variable share one reciprocal (1/x = y / (x y));
//   * riemannFlux: 1/sqrt(rho) from v_rsq_f64 serves z1, 1/z1 and the sound speeds of the entropy fix; E_t =
//     p/(gamma-1) + rho q^2/2 without the division by rho; 1/a, 1/a^2 and a from one v_rsq_f64; unit normal and area
//     from one v_rsq_f64.  8 transcendental seeds per face instead of 14 divisions + 7 square roots.
// Mapping as in kernels_inviscid_march.hip: workgroup = 64 lanes (i) x 4 rows (j) marching in k over the level's XCD-ordered
// tile table; a wavefront covers columns i0-2 .. i0+61 and produces 60 of them.
//
// gamma: calorically perfect gas (cpConstant, the only cp model of the path): gamma(i,j,k) = gammaConstant.
// Roofline: FP64 VALU issue (~1500 slots per cell), then HBM.
// The source is compiled twice (round 6): as it stands, and inside namespace adj of kernels_ad.hip with `double` standing for the dual
// number -- the exact linearisation of the adjoint (adjointUtils.F90:227-409, inviscidUpwindFlux_d) ran the cell-GATHER Roe kernel on
// dual numbers (six faces and twelve reconstructions per cell: 0.77 ms per forward pass and 1.3 M cells).  Geometry and options are
// adf_real8 and stay plain there; the dual build takes one workgroup per CU (the whole register file).
#ifndef ADF_AD_BUILD
#include "internal.h"
#endif
#include "roe_face.h"

#define RM_OUT 60          // must match EM_OUT / EM_BY of kernels_euler_march.hip: the tile table is shared
#define RM_BY 4
#ifdef ADF_AD_BUILD
#define RM_MINWG 1
#else
#define RM_MINWG 2
#endif

struct RmPtrs {
    GPTR(const double) w0; GPTR(const double) w1; GPTR(const double) w2; GPTR(const double) w3; GPTR(const double) w4;
    GPTR(const double) p;
};

__device__ __forceinline__ RCell rm_ld(const RmPtrs& m, unsigned o)
{
    RCell q;
    q.rho = ldg(m.w0, o); q.u = ldg(m.w1, o); q.v = ldg(m.w2, o); q.w = ldg(m.w3, o); q.e = ldg(m.w4, o);
    q.p = ldg(m.p, o);
    return q;
}

__device__ __forceinline__ RCell rm_up1(const RCell& q)
{
    RCell r;
    r.rho = lane_up1(q.rho); r.u = lane_up1(q.u); r.v = lane_up1(q.v); r.w = lane_up1(q.w); r.p = lane_up1(q.p); r.e = lane_up1(q.e);
    return r;
}

__device__ __forceinline__ RCell rm_dn1(const RCell& q)
{
    RCell r;
    r.rho = lane_dn1(q.rho); r.u = lane_dn1(q.u); r.v = lane_dn1(q.v); r.w = lane_dn1(q.w); r.p = lane_dn1(q.p); r.e = lane_dn1(q.e);
    return r;
}

// MUSCL states of ONE cell along one direction for one variable (leftRightState regrouped per cell, see the header):
// plus = left state of the face above the cell, minus = right state of the face below it.
template <int LIM>
__device__ __forceinline__ void rm_recon1(const RmK& K, double qm, double q0, double qp, double& plus, double& minus)
{
    if (LIM == ADFLOW_LIM_FIRST_ORDER) {          // the cell value on both faces (lumped dissipation of the preconditioner, fluxes.F90:1536)
        plus = q0; minus = q0;
        return;
    }
    const double dm = q0 - qm, dp = qp - q0;
    double A, B;
    if (LIM == ADFLOW_LIM_NONE) {
        A = dm; B = dp;
    } else {
        const double epsLim = 1.e-10;
        const double cm = copysign(fmax(fabs(dm), epsLim), dm), cp = copysign(fmax(fabs(dp), epsLim), dp);
        if (LIM == ADFLOW_LIM_VANALBADA) {
            // f(r) = r (r + 1) / (r^2 + 1) with r = max(0, a / b)  ==  a (a + b) / (a^2 + b^2) for a b > 0, else 0:
            // A = f(d+ / d-) d- = d+ d- (d+ + c-) / (d+^2 + c-^2), B likewise; both vanish unless d+ d- > 0 (one of them zero: the
            // product is the zero factor), so max(d+ d-, 0) replaces the two selects, and with the common reciprocal
            // A = s (d+ + c-) dB, B = s (d- + c+) dA, s = max(d+ d-, 0) / (dA dB): the states are q0 +- s (...)
            const double dA = dp * dp + cm * cm, dB = dm * dm + cp * cp;
            const double s = fmax(dp * dm, 0.0) * rcp_nr(dA * dB);
            const double A1 = (dp + cm) * dB, B1 = (dm + cp) * dA;
            plus = q0 + s * (K.omk * A1 + K.opk * B1);
            minus = q0 - s * (K.opk * A1 + K.omk * B1);
            return;
        } else {   // minmod: f(r) = min(1, factMinmod max(0, r))
            const double r = rcp_nr(cm * cp);
            A = fmin(1.0, K.factMinmod * fmax(0.0, dp * (r * cp))) * dm;
            B = fmin(1.0, K.factMinmod * fmax(0.0, dm * (r * cm))) * dp;
        }
    }
    plus = q0 + (K.omk * A + K.opk * B);
    minus = q0 - (K.opk * A + K.omk * B);
}

// van Albada away from the epsLim clamp: f(r) d- = f(1 / r) d+ = d+ d- (d+ + d-) / (d+^2 + d-^2) (the limiter is symmetric), and
// omk + opk = 1/2 whatever kappa is: both states are q0 +- h with h = max(d+ d-, 0) (d+ + d-) / (2 (d+^2 + d-^2)).  14 instead of
// 29 instructions per variable.  A difference that is exactly zero gives h = 0 in either form (the 1e-300 keeps 0 / 0 away; it is
// below the rounding of every other denominator); with 0 < min(|d-|, |d+|) < epsLim the clamped denominators differ and
// rm_recon1 decides: `range` collects the smallest (high word of that minimum) - 1 -- zero wraps to the top.
__device__ __forceinline__ void rm_va_sym(double qm, double q0, double qp, double& plus, double& minus, unsigned& range)
{
    const double dm = q0 - qm, dp = qp - q0;
    const double den = adf_fma(dp, dp, adf_fma(dm, dm, 1.e-300));
    const double h = (0.5 * fmax(dp * dm, 0.0)) * rcp_nr(den) * (dp + dm);
    plus = q0 + h;
    minus = q0 - h;
    const unsigned u = (unsigned)adf_hiword(fmin(fabs(dm), fabs(dp))) - 1u;
    range = (u < range) ? u : range;
}

template <int LIM>
__device__ __forceinline__ void rm_recon(const RmK& K, const RCell& a, const RCell& b, const RCell& c, double plus[5], double minus[5])
{
    if (LIM == ADFLOW_LIM_VANALBADA) {
        unsigned range = 0xffffffffu;
        rm_va_sym(a.rho, b.rho, c.rho, plus[0], minus[0], range);
        rm_va_sym(a.u, b.u, c.u, plus[1], minus[1], range);
        rm_va_sym(a.v, b.v, c.v, plus[2], minus[2], range);
        rm_va_sym(a.w, b.w, c.w, plus[3], minus[3], range);
        rm_va_sym(a.p, b.p, c.p, plus[4], minus[4], range);
#ifdef RM_COUNT_NO_CLAMP           // tools/isa_report.py: the loop as a wave executes it where no difference lies inside the clamp
        if (range < 0x3ddb7cdfu) plus[0] = 0.0;        // (keeps the range arithmetic alive: a compare + select where the kernel branches)
        return;
#endif
        if (range >= 0x3ddb7cdfu) return;      // 0x3ddb7cdf: high word of epsLim = 1e-10; an equal high word takes the clamped form too
    }
    rm_recon1<LIM>(K, a.rho, b.rho, c.rho, plus[0], minus[0]);
    rm_recon1<LIM>(K, a.u, b.u, c.u, plus[1], minus[1]);
    rm_recon1<LIM>(K, a.v, b.v, c.v, plus[2], minus[2]);
    rm_recon1<LIM>(K, a.w, b.w, c.w, plus[3], minus[3]);
    rm_recon1<LIM>(K, a.p, b.p, c.p, plus[4], minus[4]);
}

// LDS exchange of the j direction, per parity of the plane: the right states UR of the rows 0..4 of the tile (row 4 = the row
// above it, reconstructed by wave 3) and the fluxes handed to the rows 0..3 through their LOWER j face
#define RM_UR (5 * 5 * 64)
#define RM_FJ(FW) (4 * ((FW) ? 10 : 5) * 64)
#define RM_XJ(FW) (RM_UR + RM_FJ(FW))
#define RM_XQ (12 * 64)        // rows j0-1 and j0 of the plane as wave 0 holds them (six values each), for the wave of the fifth face

// FW: persistent dissipation residual of the Runge-Kutta scheme (fw kept between the stages); FINAL: dw = (dw + fw) iblank
// written here, otherwise the sum dw + fw is left in dw for the viscous kernel to complete (residuals.F90:334-344)
// ADDV (with FINAL, without FW): dw(2:5) holds the viscous flux sums of the viscous march on entry; they are added before iblank
//
// Faces per cell (round 3): a wave evaluates the k face below its cell (carried), the i face (i-1 | i) (the other one by DPP) and
// the j face ABOVE its cell; the flux through the j face BELOW comes from the wave of the row below through LDS, one plane later
// (the barrier of the next plane orders it, double-buffered by parity): the sum of cell k-1 is completed behind the barrier of
// step k.  The face below row 0 of the tile has no wave: it is the "fifth j face" and wave (k mod 4) takes it in plane k -- behind
// the barrier it takes the rows j0-1 and j0 of that plane from LDS, where wave 0 has put its own row and the row below it
// (round 4; it loaded them itself before: 17 values requested and awaited while three waves stood at the barrier), loads row
// j0-2, reconstructs cell j0-1 and hands the flux to wave 0 -- so that over four planes every SIMD carries the same load: 3.25
// face evaluations and 3.5 reconstructions per cell (4 and 3.5 before).
// RV (with FINAL, without FW): the completed residual also goes to the matrix-free residual vector kp.rvec as dw / volRef (setRVec)
template <int LIM, bool FW, bool FINAL, bool ADDV = false, bool RV = false>
__device__ __forceinline__ void roe_march_body(const BlkView* __restrict__ tab, const int4* __restrict__ tiles, const KParams& kp, int kch,
                                               int bid, double* __restrict__ xj)
{
    const int4 t = tiles[bid];
    if (t.x < 0) return;
    const BlkView& b = tab[t.x];
    const int lane = threadIdx.x, row = threadIdx.y;
    const int i = t.y * RM_OUT + lane;          // columns i0-2 .. i0+61
    const int j0 = 2 + t.z * RM_BY;
    const int j = j0 + row;
    const int k0 = 2 + t.w * kch;
    const int k1 = (k0 + kch - 1 < b.kl) ? k0 + kch - 1 : b.kl;
    const bool out = (lane >= 2 && lane <= 61 && i <= b.il && j <= b.jl);
    const int ic = (i < b.ib) ? i : b.ib;
    // rows beyond the block: clamped to the first halo row so that a valid row's upper neighbour is the true cell j+1
    const int jc = (j < b.je) ? j : b.je;
    const int jp2 = (jc + 2 < b.jb) ? jc + 2 : b.jb;
    const long nb = b.nbox;
    const unsigned sj = 8u * (unsigned)b.ldi, sk = 8u * (unsigned)b.ldk;
    unsigned c = 8u * (unsigned)(ic + jc * b.ldi + k0 * b.ldk);
    unsigned cE = 8u * (unsigned)(ic + j0 * b.ldi + k0 * b.ldk);      // row j0 of the tile (the fifth j face lies below it)
    const unsigned oj2 = 8u * (unsigned)((jp2 - jc) * b.ldi);   // offset to row j+2 (clamped)

    RmPtrs m;
    m.w0 = (GPTR(const double))b.w; m.w1 = m.w0 + nb; m.w2 = m.w1 + nb; m.w3 = m.w2 + nb; m.w4 = m.w3 + nb;
    m.p = (GPTR(const double))b.p;
    GPTR(const adf_real8) sIx = (GPTR(const adf_real8))b.sI; GPTR(const adf_real8) sIy = sIx + nb; GPTR(const adf_real8) sIz = sIy + nb;
    GPTR(const adf_real8) sJx = (GPTR(const adf_real8))b.sJ; GPTR(const adf_real8) sJy = sJx + nb; GPTR(const adf_real8) sJz = sJy + nb;
    GPTR(const adf_real8) sKx = (GPTR(const adf_real8))b.sK; GPTR(const adf_real8) sKy = sKx + nb; GPTR(const adf_real8) sKz = sKy + nb;
    GPTR(const uint8_t) flags = (GPTR(const uint8_t))b.flags;
    GPTR(double) dw = (GPTR(double))b.dw;
    GPTR(double) fw = (GPTR(double))b.fw;
    GPTR(const double) wr = (GPTR(const double))b.wr;

    RmK K;
    K.doDiss = fabs(kp.rFil) >= 1.e-10;
    K.omk = 0.25 * (1.0 - kp.kappaCoef); K.opk = 0.25 * (1.0 + kp.kappaCoef);
    K.factMinmod = (3.0 - kp.kappaCoef) / fmax(1.e-10, 1.0 - kp.kappaCoef);
    K.gam = kp.gammaConstant; K.gm1 = kp.gammaConstant - 1.0; K.ovgm1 = 1.0 / (kp.gammaConstant - 1.0);
    K.porDiss = 0.5 * kp.rFil;
    constexpr int NF = FW ? 10 : 5;             // values of one handed flux: central + dissipation apart only when fw persists

    // window k-1 .. k+1 of the own column; left state of the face above cell k-1
    RCell qm1, q0;
    double ULk[5];
    {
        const RCell qm2 = rm_ld(m, c - 2 * sk);
        qm1 = rm_ld(m, c - sk);
        q0 = rm_ld(m, c);
        double dummy[5];
        rm_recon<LIM>(K, qm2, qm1, q0, ULk, dummy);
    }
    int flagm = flags[(c - sk) >> 3];
    int flag0 = flags[c >> 3];
    adf_real8 nI[3] = {ldg(sIx, c - 8u), ldg(sIy, c - 8u), ldg(sIz, c - 8u)};     // face (i-1 | i) of the plane the step works on
    double acc[5] = {0, 0, 0, 0, 0}, accD[5] = {0, 0, 0, 0, 0};   // accD: dissipation part, kept apart only when FW

    // A wave is resident with ONE other per SIMD: what a step loads has to be in flight while it computes.  The step is laid out as
    //   request: plane k+1 of the own column, rows j-1 / j+1 of plane k, the k normal
    //   A: the i face of cell k -- state, normal and porosity are in registers since the step before
    //   B: j reconstruction + publication; request of the rows only some waves need; k reconstruction + k face; their
    //      reconstructions; request: j normal, viscous sums of cell k-1, i normal and flags of plane k+1
    //   barrier; finish cell k-1; j face (+ the fifth one)
    // with scheduling barriers between the parts (the compiler otherwise sinks every load to its first use: nine drained batches).
    // The step that only finishes the last cell stands behind the loop: inside it, its path around the j reconstruction would make
    // every wait behind that point count as if the loads of the top were still pending.
    double fc[5], fd[5];
    auto kface = [&](const RCell& qp1, adf_real8 nKx, adf_real8 nKy, adf_real8 nKz, double ULk0[5], auto&& between) {
        double URk0[5];
        rm_recon<LIM>(K, qm1, q0, qp1, ULk0, URk0);
        between();
        rm_face(K, qm1, q0, ULk, URk0, nKx, nKy, nKz, flg_porK((uint8_t)flagm), fc, fd);    // normal and porosity stored at cell k-1
    };
    // finish cell k-1 with the flux through its lower j face (handed over in the plane before) and write it
    double turbDw = 0.0;          // RV with rvecTurbFromDw: dw(itu1) of the cell finished in this step (requested with the viscous sums)
    auto finish = [&](int k, const double vsum[4]) {
        const double* __restrict__ xf = xj + ((k - 1) & 1) * RM_XJ(FW) + RM_UR;   // fluxes handed over in the plane before
        double fl[NF];
#pragma unroll
        for (int l = 0; l < NF; ++l) fl[l] = xf[(row * NF + l) * 64 + lane];
        if (out) {
            const unsigned cw = c - sk;
            const double blank = flg_blank((uint8_t)flagm);
#ifndef ADF_AD_BUILD
            double ovv = 0.0;
            double* __restrict__ rv = nullptr;
            if (RV) {
                ovv = 1.0 / ldg((GPTR(const double))b.volRef, cw);
                rv = kp.rvec + b.vecOff + ((((long)(k - 3) * b.ny + (j - 2)) * b.nx + (i - 2)) * b.nw);
            }
#endif
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                double d = (acc[l] - fl[l]) + fc[l];
                if (kp.coarseInit) d += ldg(wr + l * nb, cw);
                if (FW) {
                    double fwn = (accD[l] - fl[5 + l < NF ? 5 + l : l]) + fd[l];
                    const double old = ldg(fw + l * nb, cw);
                    fwn = K.doDiss ? (kp.sfil * old + fwn) : old;
                    if (K.doDiss || !FINAL) stg(fw + l * nb, cw, fwn);
                    stg(dw + l * nb, cw, FINAL ? (d + fwn) * blank : d);
                } else {
                    d += fd[l];
                    if (ADDV && l > 0) d += vsum[l - 1];
                    stg(dw + l * nb, cw, d * blank);      // not FINAL: the viscous kernel adds its part to dw(2:5) and re-applies iblank
#ifndef ADF_AD_BUILD
                    if (RV) rv[l] = (d * blank) * ovv;
#endif
                }
            }
#ifndef ADF_AD_BUILD
            if (RV && !FW && kp.rvecTurbFromDw) rv[5] = turbDw * ovv * kp.rvecTurbScale;
#endif
        }
    };

    for (int k = k0; k <= k1; ++k) {
        double* __restrict__ xb = xj + (k & 1) * RM_XJ(FW);                 // UR of this plane | fluxes handed over in this plane
#ifdef RM_COUNT_NO_FIFTH          // tools/isa_report.py: the loop without the block a wave executes in one plane of four
        const bool fifth = false;
#else
        const bool fifth = (wave_uniform(row) == (k & 3));                  // this wave evaluates the face below row 0 in this plane
#endif
        // ---- request
        const RCell qp1 = rm_ld(m, c + sk);
        const adf_real8 nKx = ldg(sKx, c - sk), nKy = ldg(sKy, c - sk), nKz = ldg(sKz, c - sk);
        const RCell qjm = rm_ld(m, c - sj);
        const RCell qjp = rm_ld(m, c + sj);
        RCell qjp2;
        if (row == RM_BY - 1) qjp2 = rm_ld(m, c + oj2);                     // wave 3: also reconstructs the cell above the tile
        __builtin_amdgcn_sched_barrier(0);
        // ---- A: this lane evaluates the i face (i-1 | i) of cell k; the face (i | i+1) comes from lane+1
        double gI[5], gID[5] = {0, 0, 0, 0, 0};
        {
            const RCell qL = rm_up1(q0), qR = rm_dn1(q0);
            double ULi[5], URi[5], ULm[5];
            rm_recon<LIM>(K, qL, q0, qR, ULi, URi);
#pragma unroll
            for (int l = 0; l < 5; ++l) ULm[l] = lane_up1(ULi[l]);
            const int por = flg_porI((uint8_t)lane_up1(flag0));
            double gc[5], gd[5];
            rm_face(K, qL, q0, ULm, URi, nI[0], nI[1], nI[2], por, gc, gd);
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                if (FW) {
                    gI[l] = lane_dn1(gc[l]) - gc[l];
                    gID[l] = lane_dn1(gd[l]) - gd[l];
                } else {
                    const double g = gc[l] + gd[l];
                    gI[l] = lane_dn1(g) - g;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- B: j direction, first half: reconstruct the own cell and publish the right state the row below needs.  The rows only
        //      some waves need (wave 3: the row above the tile; the wave of the fifth face: three rows below it) are requested BEHIND
        //      the wait of this reconstruction and consumed behind the k face: a conditional request in front of a wait makes the
        //      wave that issued it wait for it there
        double* __restrict__ xq = xj + 2 * RM_XJ(FW) + (k & 1) * RM_XQ;
        double ULj[5];
        {
            double URj[5];
            rm_recon<LIM>(K, qjm, q0, qjp, ULj, URj);
#pragma unroll
            for (int l = 0; l < 5; ++l) xb[(row * 5 + l) * 64 + lane] = URj[l];
            if (row == 0) {               // rows j0-1 and j0 of this plane for the wave of the fifth face
                xq[0 * 64 + lane] = qjm.rho; xq[1 * 64 + lane] = qjm.u; xq[2 * 64 + lane] = qjm.v; xq[3 * 64 + lane] = qjm.w;
                xq[4 * 64 + lane] = qjm.p; xq[5 * 64 + lane] = qjm.e;
                xq[6 * 64 + lane] = q0.rho; xq[7 * 64 + lane] = q0.u; xq[8 * 64 + lane] = q0.v; xq[9 * 64 + lane] = q0.w;
                xq[10 * 64 + lane] = q0.p; xq[11 * 64 + lane] = q0.e;
            }
        }
        double ULk0[5];
        kface(qp1, nKx, nKy, nKz, ULk0, []() {});
        __builtin_amdgcn_sched_barrier(0);
        if (row == RM_BY - 1) {                                          // wave 3: also the cell above the tile
            double pl[5], mi[5];
            rm_recon<LIM>(K, q0, qjp, qjp2, pl, mi);
#pragma unroll
            for (int l = 0; l < 5; ++l) xb[(4 * 5 + l) * 64 + lane] = mi[l];
        }
        // request what the part behind the barrier and phase A of the next step consume
        adf_real8 nJ[3], nIn[3], nE[3];
        double vsum[4] = {0, 0, 0, 0};       // ADDV: viscous flux sums of the cell finished in this step
        if (ADDV && k > k0 && out) {
#pragma unroll
            for (int l = 0; l < 4; ++l) vsum[l] = ldg((GPTR(const double))dw + (l + 1) * nb, c - sk);
        }
        if (RV && kp.rvecTurbFromDw && k > k0 && out) turbDw = ldg((GPTR(const double))dw + 5 * nb, c - sk);
        nJ[0] = ldg(sJx, c); nJ[1] = ldg(sJy, c); nJ[2] = ldg(sJz, c);
        nIn[0] = ldg(sIx, c + sk - 8u); nIn[1] = ldg(sIy, c + sk - 8u); nIn[2] = ldg(sIz, c + sk - 8u);
        const int flagp = flags[(c + sk) >> 3];
        // (normal and flags of the fifth face by every wave: inside the condition the compiler decodes the flags there and then, i.e.
        // makes that wave wait for this whole request in front of the barrier)
        RCell qEmm;
        const unsigned cm = cE - sj;
        nE[0] = ldg(sJx, cm); nE[1] = ldg(sJy, cm); nE[2] = ldg(sJz, cm);
        const int flagE = flags[cm >> 3];
        if (fifth) qEmm = rm_ld(m, cE - 2 * sj);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        if (k > k0) finish(k, vsum);
        // ---- start cell k: the k face below it and its two i faces
#pragma unroll
        for (int l = 0; l < 5; ++l) {
            ULk[l] = ULk0[l];
            if (FW) { acc[l] = gI[l] - fc[l]; accD[l] = gID[l] - fd[l]; }
            else acc[l] = gI[l] - (fc[l] + fd[l]);
        }
        // ---- j face above the cell with the right state of the row above; its flux is handed to that row
        {
            double Rp[5];
#pragma unroll
            for (int l = 0; l < 5; ++l) Rp[l] = xb[((row + 1) * 5 + l) * 64 + lane];
            double hc[5], hd[5];
            rm_face(K, q0, qjp, ULj, Rp, nJ[0], nJ[1], nJ[2], flg_porJ((uint8_t)flag0), hc, hd);
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                if (FW) { acc[l] += hc[l]; accD[l] += hd[l]; }
                else acc[l] += hc[l] + hd[l];
            }
            if (row < RM_BY - 1) {
                double* __restrict__ fo = xb + RM_UR + ((row + 1) * NF) * 64 + lane;
#pragma unroll
                for (int l = 0; l < 5; ++l) {
                    if (FW) { fo[l * 64] = hc[l]; fo[(5 + l) * 64] = hd[l]; }
                    else fo[l * 64] = hc[l] + hd[l];
                }
            }
        }
        // ---- the fifth j face (j0-1 | j0): right state = UR of row 0
        if (fifth) {
            RCell qEm, qE0;
            qEm.rho = xq[0 * 64 + lane]; qEm.u = xq[1 * 64 + lane]; qEm.v = xq[2 * 64 + lane]; qEm.w = xq[3 * 64 + lane];
            qEm.p = xq[4 * 64 + lane]; qEm.e = xq[5 * 64 + lane];
            qE0.rho = xq[6 * 64 + lane]; qE0.u = xq[7 * 64 + lane]; qE0.v = xq[8 * 64 + lane]; qE0.w = xq[9 * 64 + lane];
            qE0.p = xq[10 * 64 + lane]; qE0.e = xq[11 * 64 + lane];
            double ULe[5], mi[5];
            rm_recon<LIM>(K, qEmm, qEm, qE0, ULe, mi);                  // left state of the face (j0-1 | j0)
            double Rm[5];
#pragma unroll
            for (int l = 0; l < 5; ++l) Rm[l] = xb[l * 64 + lane];
            double hc[5], hd[5];
            rm_face(K, qEm, qE0, ULe, Rm, nE[0], nE[1], nE[2], flg_porJ((uint8_t)flagE), hc, hd);
            double* __restrict__ fo = xb + RM_UR + lane;
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                if (FW) { fo[l * 64] = hc[l]; fo[(5 + l) * 64] = hd[l]; }
                else fo[l * 64] = hc[l] + hd[l];
            }
        }
        // ---- advance the window
        qm1 = q0; q0 = qp1;
        flagm = flag0; flag0 = flagp;
#pragma unroll
        for (int d = 0; d < 3; ++d) nI[d] = nIn[d];
        c += sk; cE += sk;
    }
    // ---- the k face above the last cell of the chunk, and that cell
    {
        const RCell qp1 = rm_ld(m, c + sk);
        const adf_real8 nKx = ldg(sKx, c - sk), nKy = ldg(sKy, c - sk), nKz = ldg(sKz, c - sk);
        double vsum[4] = {0, 0, 0, 0};
        if (ADDV && out) {
#pragma unroll
            for (int l = 0; l < 4; ++l) vsum[l] = ldg((GPTR(const double))dw + (l + 1) * nb, c - sk);
        }
        if (RV && kp.rvecTurbFromDw && out) turbDw = ldg((GPTR(const double))dw + 5 * nb, c - sk);
        double ULk0[5];
        kface(qp1, nKx, nKy, nKz, ULk0, []() {});
        __syncthreads();
        finish(k1 + 1, vsum);
    }
}

template <int LIM, bool FW, bool FINAL, bool ADDV = false, bool RV = false>
__global__ __launch_bounds__(64 * RM_BY, RM_MINWG) void k_roe_march(const BlkView* __restrict__ tab, const int4* __restrict__ tiles, KParams kp,
                                                             int kch)
{
    __shared__ double xj[2 * RM_XJ(FW) + 2 * RM_XQ];
    roe_march_body<LIM, FW, FINAL, ADDV, RV>(tab, tiles, kp, kch, (int)blockIdx.x, xj);
}

#ifndef ADF_AD_BUILD
int g_roe_march = 1;       // tuning "roe_march": 0 = k_inviscid_march<upwind> (reconstruction per face) on the fine level too

extern int g_march_kch;
#endif

template <int LIM>
static void launch_rm(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s)
{
    const dim3 blk(64, RM_BY, 1), grd(ntiles);
    const bool doDiss = fabs(kp.rFil) >= 1.e-10;
    const bool final_ = !(kp.viscous && doDiss);
    const int kch = ::g_march_kch;
#ifdef ADF_AD_BUILD
    // forward mode (the exact dR/dw of the adjoint): no persistent fw, no matrix-free vector; the dual gather viscous kernel follows
    // and completes dw where the equations are viscous
    // (viscFirst: the dual form of k_visc_gf ran in front and left its flux sums in dw(2:5))
    if (kp.viscFirst) hipLaunchKernelGGL((k_roe_march<LIM, false, true, true>), grd, blk, 0, s, tab, tiles, kp, kch);
    else if (final_) hipLaunchKernelGGL((k_roe_march<LIM, false, true>), grd, blk, 0, s, tab, tiles, kp, kch);
    else hipLaunchKernelGGL((k_roe_march<LIM, false, false>), grd, blk, 0, s, tab, tiles, kp, kch);
#else
    if (kp.fwMode) {
        if (final_) hipLaunchKernelGGL((k_roe_march<LIM, true, true>), grd, blk, 0, s, tab, tiles, kp, kch);
        else hipLaunchKernelGGL((k_roe_march<LIM, true, false>), grd, blk, 0, s, tab, tiles, kp, kch);
    } else if (kp.viscFirst) {
        if (kp.rvec) {
            hipLaunchKernelGGL((k_roe_march<LIM, false, true, true, true>), grd, blk, 0, s, tab, tiles, kp, kch);
            adf_note_rvec(kp.rvecTurbFromDw ? 3 : 1);
        } else
            hipLaunchKernelGGL((k_roe_march<LIM, false, true, true>), grd, blk, 0, s, tab, tiles, kp, kch);
    } else {
        if (final_ && kp.rvec) {
            hipLaunchKernelGGL((k_roe_march<LIM, false, true, false, true>), grd, blk, 0, s, tab, tiles, kp, kch);
            adf_note_rvec(kp.rvecTurbFromDw ? 3 : 1);
        } else if (final_) hipLaunchKernelGGL((k_roe_march<LIM, false, true>), grd, blk, 0, s, tab, tiles, kp, kch);
        else hipLaunchKernelGGL((k_roe_march<LIM, false, false>), grd, blk, 0, s, tab, tiles, kp, kch);
    }
#endif
}

bool roe_march_takes(const KParams& kp)
{
    if (!::g_roe_march || kp.spaceDiscr != ADFLOW_UPWIND || !kp.fineGrid) return false;
    const int lim = kp.lumpedDiss ? ADFLOW_LIM_FIRST_ORDER : kp.limiter;
    return lim == ADFLOW_LIM_FIRST_ORDER || lim == ADFLOW_LIM_NONE || lim == ADFLOW_LIM_VANALBADA || lim == ADFLOW_LIM_MINMOD;
}

// true when the launch was taken: second-order Roe upwind on the fine level of blocks at rest
bool launch_roe_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s)
{
    // (the approximate residual changes the Roe scheme only through the limiter: lumpedDiss = first order)
    if (!::g_roe_march || kp.spaceDiscr != ADFLOW_UPWIND || !kp.fineGrid) return false;
    if (ntiles <= 0) return true;
    switch (kp.lumpedDiss ? ADFLOW_LIM_FIRST_ORDER : kp.limiter) {
    case ADFLOW_LIM_FIRST_ORDER: launch_rm<ADFLOW_LIM_FIRST_ORDER>(tab, tiles, ntiles, kp, s); return true;
    case ADFLOW_LIM_NONE: launch_rm<ADFLOW_LIM_NONE>(tab, tiles, ntiles, kp, s); return true;
    case ADFLOW_LIM_VANALBADA: launch_rm<ADFLOW_LIM_VANALBADA>(tab, tiles, ntiles, kp, s); return true;
    case ADFLOW_LIM_MINMOD: launch_rm<ADFLOW_LIM_MINMOD>(tab, tiles, ntiles, kp, s); return true;
    default: return false;
    }
}
