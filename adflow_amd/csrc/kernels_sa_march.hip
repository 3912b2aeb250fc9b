// The Spalart-Allmaras residual as a k-march (moved out of kernels_viscous.hip in round 5).  The source is compiled twice: as it
// stands, and inside namespace adj of kernels_ad.hip with `double` standing for the dual number of dual.h (the turbulence residual of
// the forward-mode preconditioner matrix: the cell-gather twin cost 1.6 ms per evaluation on the north-star mesh) -- geometry,
// wall distance and options are typed adf_real8 and stay plain there; offsets are byte offsets of 8-byte elements, which the dual
// forms of ldg / stg double (kernels_ad.hip).
//   saSource        src/turbulence/sa.F90:89-344
//   turbAdvection   src/turbulence/turbUtils.F90:828-1561
//   saViscous       src/turbulence/sa.F90:346-676
#ifndef ADF_AD_BUILD
#include "internal.h"
#endif
#include "sa_core.h"

#define GS_BY 4            // rows of a tile (the chunk table of ensure_sa_tiles, api.hip)
#ifdef ADF_AD_BUILD
#define GS_MINWG 1         // dual numbers: the whole register file (352 B of scratch per lane at two workgroups per CU)
#else
#define GS_MINWG 2
#endif

__device__ __forceinline__ void gs_ld3(GPTR(const adf_real8) a, unsigned o, unsigned nb8, adf_real8 v[3])
{
    v[0] = ldg(a, o); v[1] = ldg(a, o + nb8); v[2] = ldg(a, o + 2 * nb8);
}

// ---------------------------------------------------------------------------
// The Spalart-Allmaras residual as a k-march.  k_sa_residual gathers 73 values per cell (237 B per cell from HBM, bound by load
// latency).  Here a thread keeps a three-plane window of its own column (u, v, w, nu = rlv / rho, vol; five planes of nuTilde),
// takes the i neighbours by DPP lane shifts and loads only the j neighbours, d2Wall and volRef; the face normals are re-formed
// from the node coordinates (tuning metric_from_x bit 0) or loaded.  No LDS, no barrier.
// Tiles advance by 60: cells of lanes 2..61 (second-order advection reaches i +- 2).
// Work list: the level's round-fitted chunk table (api.hip ensure_sa_tiles; x = slot or -1, y = bx | by << 16, z / w = first / last cell
// plane): 60 columns x 4 rows x (k0 .. k1), one loop trip per produced plane.
// Arithmetic of the SA terms: sa_core.h (shared with k_sa_residual), same order of the sweeps (k, j, i).
// ---------------------------------------------------------------------------
#define GS_OUT 60



struct GsCell { double u, v, w, rho, rlv; adf_real8 vol; };      // own column, one plane
struct GsNbr { double u, v, w, nu, nut; adf_real8 vol; };           // what a neighbour contributes to the SA terms

struct GsPtrs {
    GPTR(const double) w0; GPTR(const double) w1; GPTR(const double) w2; GPTR(const double) w3; GPTR(const double) w5;
    GPTR(const double) rlv; GPTR(const adf_real8) vol;
    GPTR(const adf_real8) sI; GPTR(const adf_real8) sJ; GPTR(const adf_real8) sK;
    GPTR(const adf_real8) d2wall; GPTR(const adf_real8) volRef;
    unsigned nb8, sj;
};

__device__ __forceinline__ GsCell gs_ld(const GsPtrs& m, unsigned c)
{
    GsCell q;
    q.rho = ldg(m.w0, c); q.u = ldg(m.w1, c); q.v = ldg(m.w2, c); q.w = ldg(m.w3, c);
    q.vol = ldg(m.vol, c); q.rlv = ldg(m.rlv, c);
    return q;
}

__device__ __forceinline__ GsNbr gs_nbr_ld(const GsPtrs& m, unsigned c)
{
    GsNbr q;
    q.u = ldg(m.w1, c); q.v = ldg(m.w2, c); q.w = ldg(m.w3, c);
    q.nu = ldg(m.rlv, c) * rcp_nr(ldg(m.w0, c));
    q.vol = ldg(m.vol, c);
    q.nut = ldg(m.w5, c);
    return q;
}

// the same in two halves: request (seven values in flight), then the kinematic viscosity
struct GsNbrRaw { double u, v, w, rlv, rho, nut; adf_real8 vol; };
__device__ __forceinline__ GsNbrRaw gs_nbr_req(const GsPtrs& m, unsigned c)
{
    GsNbrRaw q;
    q.u = ldg(m.w1, c); q.v = ldg(m.w2, c); q.w = ldg(m.w3, c);
    q.rlv = ldg(m.rlv, c); q.rho = ldg(m.w0, c);
    q.vol = ldg(m.vol, c);
    q.nut = ldg(m.w5, c);
    return q;
}
__device__ __forceinline__ GsNbr gs_nbr_of(const GsNbrRaw& r)
{
    GsNbr q;
    q.u = r.u; q.v = r.v; q.w = r.w; q.nu = r.rlv * rcp_nr(r.rho); q.vol = r.vol; q.nut = r.nut;
    return q;
}

__device__ __forceinline__ GsNbr gs_up1(const GsNbr& q)
{
    GsNbr r;
    r.u = lane_up1(q.u); r.v = lane_up1(q.v); r.w = lane_up1(q.w); r.nu = lane_up1(q.nu); r.vol = lane_up1(q.vol); r.nut = lane_up1(q.nut);
    return r;
}
__device__ __forceinline__ GsNbr gs_dn1(const GsNbr& q)
{
    GsNbr r;
    r.u = lane_dn1(q.u); r.v = lane_dn1(q.v); r.w = lane_dn1(q.w); r.nu = lane_dn1(q.nu); r.vol = lane_dn1(q.vol); r.nut = lane_dn1(q.nut);
    return r;
}

// SOLVE (saSolve): also stores the right-hand side (scratch 0) and the central jacobian qq (scratch 1) of the DDADI line solves,
// as k_sa_residual<true> (kernels_sa.hip)
// RV: the residual also goes to the matrix-free residual vector kp.rvec (setRVec: dw / volRef * turbResScale)
// SNAP: the Jacobian assembly's snapshot entry instead of dw (KParams::snapTab; compile-time: in the dual build the value-only
// arithmetic behind the stored derivative goes away)
template <bool SOLVE, bool RV = false, bool SNAP = false>
__global__ __launch_bounds__(64 * GS_BY, GS_MINWG) void k_sa_march(const BlkView* __restrict__ tab, const int4* __restrict__ tiles, KParams kp)
{
    const int4 tl = tiles[blockIdx.x];
    if (tl.x < 0) return;
    const BlkView& b = tab[tl.x];
    const int lane = threadIdx.x, row = threadIdx.y;
    const int kn0 = tl.z, kn1 = tl.w;                          // cell planes of the chunk (2 .. kl)
    const int i = (tl.y & 0xffff) * GS_OUT + lane, j = 2 + (tl.y >> 16) * GS_BY + row;      // cells of lanes 2 .. 61
    const int ic = (i < b.ib) ? i : b.ib, jc = (j < b.jb) ? j : b.jb;
    const bool outC = (lane >= 2 && lane <= GS_OUT + 1 && i <= b.il && j <= b.jl);   // SA cell produced
    const long nb = b.nbox;
    GsPtrs m;
    m.w0 = (GPTR(const double))b.w; m.w1 = m.w0 + nb; m.w2 = m.w1 + nb; m.w3 = m.w2 + nb; m.w5 = m.w3 + 2 * nb;
    m.rlv = (GPTR(const double))b.rlv; m.vol = (GPTR(const adf_real8))b.vol;
    m.sI = (GPTR(const adf_real8))b.sI; m.sJ = (GPTR(const adf_real8))b.sJ; m.sK = (GPTR(const adf_real8))b.sK;
    m.d2wall = (GPTR(const adf_real8))b.d2wall; m.volRef = (GPTR(const adf_real8))b.volRef;
    m.nb8 = 8u * (unsigned)nb; m.sj = 8u * (unsigned)b.ldi;
    GPTR(const adf_real8) xnod = (GPTR(const adf_real8))b.x;
    const adf_real8 mfact = b.mfact;
    const int xn = (kp.metricFromX & 1);                  // face normals re-formed from the node coordinates
    GPTR(double) dw5 = (GPTR(double))b.dw + 5 * nb;
    GPTR(const uint8_t) flags = (GPTR(const uint8_t))b.flags;
    const unsigned sk = 8u * (unsigned)b.ldk, sj = m.sj;
    // rows j +- 2 of nuTilde, clamped into the box (only read for produced cells, where they are inside)
    const unsigned ojm2 = (jc >= 2) ? 2 * sj : (unsigned)jc * sj, ojp2 = (jc + 2 <= b.jb) ? 2 * sj : (unsigned)(b.jb - jc) * sj;
    const unsigned ojm1 = (jc >= 1) ? sj : 0u, ojp1 = (jc + 1 <= b.jb) ? sj : 0u;
    unsigned c = 8u * (unsigned)(ic + jc * b.ldi + kn0 * b.ldk);
    const bool secondOrd = (kp.orderTurb == 2) && kp.groundLevelIsOne;
    const adf_real8 cb3Inv = 1.0 / kp.sa_cb3;
    adf_real8 sKp[3];
    NgNodes Pn;
    if (xn) {
        ngx_load_x(xnod, c - sk, m.nb8, m.sj, Pn);
        ngx_normal_k(mfact, Pn, sKp);
    } else {
#pragma unroll
        for (int d = 0; d < 3; ++d) sKp[d] = ldg(m.sK, c - sk + d * m.nb8);
    }
    // window of the own column: planes kn0-1 (SA neighbour below) and kn0; nuTilde of planes kn0-2 .. kn0+1
    GsCell s0 = gs_ld(m, c);
    GsNbr sm1 = gs_nbr_ld(m, c - sk);
    double n_m2 = ldg(m.w5, c - 2 * sk), n_0 = ldg(m.w5, c), n_p1 = ldg(m.w5, c + sk);
    for (int mm = kn0; mm <= kn1; ++mm) {
        const unsigned ckp2 = (mm + 2 <= b.kb) ? 2 * sk : ((mm + 1 <= b.kb) ? sk : 0u);
        const unsigned ckp1 = (mm + 1 <= b.kb) ? sk : 0u;
        // ---- loads of this plane, ONE batch in front of the arithmetic (the wave shares its SIMD with one other: every further
        //      batch is a further exposed latency): the state of the plane above, nuTilde two planes above, the j neighbours, wall
        //      distance, reference volume, flags; then the four face-normal triples (or the node rows they are formed from)
        const GsCell sp1 = gs_ld(m, c + ckp1);
        const double n_p2 = ldg(m.w5, c + ckp2);
        const GsNbrRaw rjm = gs_nbr_req(m, c - ojm1), rjp = gs_nbr_req(m, c + ojp1);
        const double n_jm2 = ldg(m.w5, c - ojm2), n_jp2 = ldg(m.w5, c + ojp2);
        const adf_real8 d2w = ldg(m.d2wall, c), volRef0 = ldg(m.volRef, c);
        const int flag0 = flags[c >> 3];
        adf_real8 nI[3], nJm[3], nJ[3], nK[3];
        if (xn) {
            NgNodes Nn;
            ngx_load_x(xnod, c, m.nb8, m.sj, Nn);
            ngx_normals(mfact, Pn, Nn, nI, nJm, nJ, nK);
            Pn = Nn;
        } else {
            gs_ld3(m.sI, c, m.nb8, nI); gs_ld3(m.sJ, c - ojm1, m.nb8, nJm); gs_ld3(m.sJ, c, m.nb8, nJ); gs_ld3(m.sK, c, m.nb8, nK);
        }
        __builtin_amdgcn_sched_barrier(0);
        const adf_real8 sKm[3] = {sKp[0], sKp[1], sKp[2]};    // sK of the plane below
#pragma unroll
        for (int d = 0; d < 3; ++d) sKp[d] = nK[d];
        // ---- Spalart-Allmaras residual of cell (i, j, mm): sweeps k, j, i as the reference (sa.F90, turbUtils.F90)
        {
            GsNbr q0;
            q0.u = s0.u; q0.v = s0.v; q0.w = s0.w; q0.nu = s0.rlv * rcp_nr(s0.rho); q0.vol = s0.vol; q0.nut = n_0;
            GsNbr qkp;
            qkp.u = sp1.u; qkp.v = sp1.v; qkp.w = sp1.w; qkp.nu = sp1.rlv * rcp_nr(sp1.rho); qkp.vol = sp1.vol; qkp.nut = n_p1;
            const GsNbr qim = gs_up1(q0), qip = gs_dn1(q0);
            const double n_im2 = lane_up1(qim.nut), n_ip2 = lane_dn1(qip.nut);
            const GsNbr qjm = gs_nbr_of(rjm), qjp = gs_nbr_of(rjp);
            const adf_real8 nIm[3] = {lane_up1(nI[0]), lane_up1(nI[1]), lane_up1(nI[2])};
            // velocity gradient * 2 vol from the six neighbours (sa.F90:133-190)
            double gu[3][3];
            const double qq[3][6] = {{qip.u, qim.u, qjp.u, qjm.u, qkp.u, sm1.u}, {qip.v, qim.v, qjp.v, qjm.v, qkp.v, sm1.v},
                                     {qip.w, qim.w, qjp.w, qjm.w, qkp.w, sm1.w}};
#pragma unroll
            for (int v = 0; v < 3; ++v)
#pragma unroll
                for (int d = 0; d < 3; ++d)
                    gu[v][d] = qq[v][0] * nI[d] - qq[v][1] * nIm[d] + qq[v][2] * nJ[d] - qq[v][3] * nJm[d] + qq[v][4] * nK[d] - qq[v][5] * sKm[d];
            double qjac = 0.0;
            double dvt = sa_source(kp, gu, s0.vol, q0.nu, n_0, d2w, SOLVE ? &qjac : nullptr);
            SaDir dk, dj, di;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                dk.sm[d] = sKm[d]; dk.sp[d] = nK[d]; dj.sm[d] = nJm[d]; dj.sp[d] = nJ[d]; di.sm[d] = nIm[d]; di.sp[d] = nI[d];
            }
            dk.volm = sm1.vol; dk.volp = qkp.vol; dk.num = sm1.nu; dk.nup = qkp.nu; dk.qsf = 0.0;
            dk.nt[0] = n_m2; dk.nt[1] = sm1.nut; dk.nt[2] = n_0; dk.nt[3] = n_p1; dk.nt[4] = n_p2;
            dj.volm = qjm.vol; dj.volp = qjp.vol; dj.num = qjm.nu; dj.nup = qjp.nu; dj.qsf = 0.0;
            dj.nt[0] = n_jm2; dj.nt[1] = qjm.nut; dj.nt[2] = n_0; dj.nt[3] = qjp.nut; dj.nt[4] = n_jp2;
            di.volm = qim.vol; di.volp = qip.vol; di.num = qim.nu; di.nup = qip.nu; di.qsf = 0.0;
            di.nt[0] = n_im2; di.nt[1] = qim.nut; di.nt[2] = n_0; di.nt[3] = qip.nut; di.nt[4] = n_ip2;
            if (SOLVE) {
                // central jacobian of advection and diffusion with the implicit boundary part (turbUtils.F90:972-1004,1060-1092,
                // sa.F90:452-468): max(bmt, 0) of the face behind a boundary cell
                double bmK1 = 0.0, bmK2 = 0.0, bmJ1 = 0.0, bmJ2 = 0.0, bmI1 = 0.0, bmI2 = 0.0;
                if (b.bmt[0] && outC) {
                    if (i == 2) bmI1 = fmax(b.bmt[0][(j - 1) + (long)b.je * (mm - 1)], 0.0);
                    if (i == b.il) bmI2 = fmax(b.bmt[1][(j - 1) + (long)b.je * (mm - 1)], 0.0);
                    if (j == 2) bmJ1 = fmax(b.bmt[2][(i - 1) + (long)b.ie * (mm - 1)], 0.0);
                    if (j == b.jl) bmJ2 = fmax(b.bmt[3][(i - 1) + (long)b.ie * (mm - 1)], 0.0);
                    if (mm == 2) bmK1 = fmax(b.bmt[4][(i - 1) + (long)b.ie * (j - 1)], 0.0);
                    if (mm == b.kl) bmK2 = fmax(b.bmt[5][(i - 1) + (long)b.ie * (j - 1)], 0.0);
                }
                // the off-diagonals of the three line solves of saSolve (sa.F90:858-1240: bb = (-c1m - max(uu, 0)) rblank, dd = (-c1p +
                // min(uu, 0)) rblank) depend on the frozen state only and everything they are made of is at hand here: written once
                // (scratch 3 .. 8: bb, dd of j, i, k), the sweeps then read two values per cell and direction instead of forming them
                // from ~20
                double uuK, uuJ, uuI, c1m, c1p;
                const adf_real8 rbl = flg_blank((uint8_t)flag0);
                GPTR(double) scr = (GPTR(double))b.scratch;
                dvt += sa_advect(dk, s0.vol, s0.u, s0.v, s0.w, secondOrd, &uuK); qjac += fabs(uuK) + ((uuK > 0.0) ? uuK * bmK1 : -uuK * bmK2);
                dvt += sa_advect(dj, s0.vol, s0.u, s0.v, s0.w, secondOrd, &uuJ); qjac += fabs(uuJ) + ((uuJ > 0.0) ? uuJ * bmJ1 : -uuJ * bmJ2);
                dvt += sa_advect(di, s0.vol, s0.u, s0.v, s0.w, secondOrd, &uuI); qjac += fabs(uuI) + ((uuI > 0.0) ? uuI * bmI1 : -uuI * bmI2);
                dvt += sa_diffuse(dk, s0.vol, q0.nu, kp.sa_cb2, cb3Inv, &c1m, &c1p); qjac += c1m + c1p + ((mm == 2) ? c1m * bmK1 : c1p * bmK2);
                if (outC) { stg(scr + 7 * nb, c, (-c1m - fmax(uuK, 0.0)) * rbl); stg(scr + 8 * nb, c, (-c1p + fmin(uuK, 0.0)) * rbl); }
                dvt += sa_diffuse(dj, s0.vol, q0.nu, kp.sa_cb2, cb3Inv, &c1m, &c1p); qjac += c1m + c1p + ((j == 2) ? c1m * bmJ1 : c1p * bmJ2);
                if (outC) { stg(scr + 3 * nb, c, (-c1m - fmax(uuJ, 0.0)) * rbl); stg(scr + 4 * nb, c, (-c1p + fmin(uuJ, 0.0)) * rbl); }
                dvt += sa_diffuse(di, s0.vol, q0.nu, kp.sa_cb2, cb3Inv, &c1m, &c1p); qjac += c1m + c1p + ((i == 2) ? c1m * bmI1 : c1p * bmI2);
                if (outC) { stg(scr + 5 * nb, c, (-c1m - fmax(uuI, 0.0)) * rbl); stg(scr + 6 * nb, c, (-c1p + fmin(uuI, 0.0)) * rbl); }
                if (outC) {
                    stg(scr, c, dvt);
                    stg(scr + nb, c, kp.sa_qqFactor * qjac);      // implicit relaxation factor of saSolve (sa.F90:830-836)
                }
            } else {
                dvt += sa_advect(dk, s0.vol, s0.u, s0.v, s0.w, secondOrd);
                dvt += sa_advect(dj, s0.vol, s0.u, s0.v, s0.w, secondOrd);
                dvt += sa_advect(di, s0.vol, s0.u, s0.v, s0.w, secondOrd);
                dvt += sa_diffuse(dk, s0.vol, q0.nu, kp.sa_cb2, cb3Inv);
                dvt += sa_diffuse(dj, s0.vol, q0.nu, kp.sa_cb2, cb3Inv);
                dvt += sa_diffuse(di, s0.vol, q0.nu, kp.sa_cb2, cb3Inv);
            }
            if (outC) {
                const adf_real8 blank = flg_blank((uint8_t)flag0);
                const double d5 = -volRef0 * dvt * blank;
                if (SNAP) {
                    // Jacobian assembly: resScale + the snapshot entry of this coloured evaluation instead of dw (KParams::snapTab)
                    const SnapSlot ss = kp.snapTab[tl.x];
                    const long m = (long)kp.snapCol * kp.snapN + (5 - kp.snapL0);
                    snap_put((GPTR(adf_real8))ss.snap + m * nb, c, d5 * (1.0 / volRef0) * kp.snapTurbScale);
                } else
                    stg(dw5, c, d5);
                // setRVec of the matrix-free matvec: dw / volRef * turbResScale
#ifndef ADF_AD_BUILD
                if (RV)
                    kp.rvec[b.vecOff + ((((long)(mm - 2) * b.ny + (j - 2)) * b.nx + (i - 2)) * b.nw) + 5] = -dvt * blank * kp.rvecTurbScale;
#endif
            }
        }
        // ---- advance the window
        n_m2 = sm1.nut;
        sm1.u = s0.u; sm1.v = s0.v; sm1.w = s0.w; sm1.nu = s0.rlv * rcp_nr(s0.rho); sm1.vol = s0.vol; sm1.nut = n_0;
        s0 = sp1;
        n_0 = n_p1; n_p1 = n_p2;
        c += ckp1;
    }
}


// the Spalart-Allmaras residual as a k-march over a chunk table of the level (blocks at rest); solve: also the right-hand side and the
// central jacobian of saSolve
void launch_sa_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s, bool solve)
{
    if (ntiles <= 0) return;
    const dim3 grd(ntiles), blk(64, GS_BY, 1);
#ifndef ADF_AD_BUILD
    if (solve) hipLaunchKernelGGL((k_sa_march<true>), grd, blk, 0, s, tab, tiles, kp);
    else if (kp.rvec && !kp.rvecTurbFromDw) {
        hipLaunchKernelGGL((k_sa_march<false, true>), grd, blk, 0, s, tab, tiles, kp);
        adf_note_rvec(2);
    } else
#endif
        if (kp.snapTab) { hipLaunchKernelGGL((k_sa_march<false, false, true>), grd, blk, 0, s, tab, tiles, kp); adf_note_snap(2); }
        else hipLaunchKernelGGL((k_sa_march<false>), grd, blk, 0, s, tab, tiles, kp);     // (the dual build: the residual alone, or its snapshot)
}

