// The mean-flow residual of the PRECONDITIONER MATRIX as one k-marching kernel: central flux + FIRST-ORDER Roe flux + thin-layer
// viscous flux, written once.  This is what one coloured evaluation of adjointUtils::setupStateResidualMatrix executes for the
// flow equations when usePC = T on the upwind scheme (adjointUtils.F90:176-191: lumpedDiss = T -> first-order upwind
// (fluxes.F90:1536), viscApprox -> viscousFluxApprox (fluxes.F90:3487-3859)); masterRoutines::block_res_state :1269-1277 (finite
// differences) and block_res_state_d :1363-1385 (forward mode).
//
// Why a kernel of its own (round 5): the finite-difference assembly ran k_visc_approx_march (0.61 ms: state + 18 geometry arrays
// in, four sums out) and k_roe_march<first order> (0.50 ms: the state again, 9 geometry arrays, the four sums back in) -- with
// first-order states BOTH fluxes of a face are functions of the two cells beside it, so one pass over the state serves both; the
// forward-mode assembly ran the cell-GATHER kernels on dual numbers (six Roe faces + six viscous faces per cell, 3.4 + 1.9 ms).
// The source below is compiled twice: as it stands, and inside namespace adj of kernels_ad.hip with `double` standing for the dual
// number (dual.h) -- state, fluxes and residual carry value + derivative, geometry and options (adf_real8) stay plain.
//
// Mapping: the level's tile table (60 of 64 lanes produce, 4 rows, chunks of kch planes).  k face carried, i face once (its flux
// comes back from the neighbouring lane by DPP); the states of the neighbouring rows through LDS (double buffered by the parity of
// the plane: one barrier per plane), the rows outside the tile loaded by the waves next to them.  Every j face ONCE (the flux handed to
// the row above, the cell completed a plane later: 3.25 faces per cell; see there).  SNAP: the result goes to the snapshot of the
// Jacobian sweep.
// Measured (north-star mesh, profiles/r05_fin6_pc_pmc_bytes.md, r05_fin4_pc_trace.md): plain 0.70 ms, 329 B per cell counted =
// 4.9 TB/s; dual 1.52 ms (one wave per SIMD, bound by FP64 issue).
#ifndef ADF_AD_BUILD
#include "internal.h"
#endif
#include "roe_face.h"

#define PM_OUT 60          // tile table shared with the other marching kernels
#define PM_BY 4
#define PM_NV 9
#ifdef ADF_AD_BUILD
#define PM_MINWG 1         // dual numbers: the whole register file (at two workgroups per CU the kernel spills 750 B per lane)
#else
#define PM_MINWG 2
#endif

struct PcCell { double rho, u, v, w, p, e, na, rlv, rev; };      // na = - gamma p / rho (minus the speed of sound squared)

struct PcPtrs {
    GPTR(const double) w0; GPTR(const double) w1; GPTR(const double) w2; GPTR(const double) w3; GPTR(const double) w4;
    GPTR(const double) p; GPTR(const double) rlv; GPTR(const double) rev;
};

struct PcK { adf_real8 porV, hl, ht, gam; bool eddy; };           // 0.5 rFil; 1 / (prandtl (gamma-1)); 1 / (prandtlTurb (gamma-1))

__device__ __forceinline__ PcCell pc_ld(const PcPtrs& m, unsigned o, const PcK& V)
{
    PcCell q;
    q.rho = ldg(m.w0, o); q.u = ldg(m.w1, o); q.v = ldg(m.w2, o); q.w = ldg(m.w3, o); q.e = ldg(m.w4, o);
    q.p = ldg(m.p, o);
    q.rlv = ldg(m.rlv, o);
    if (V.eddy) q.rev = ldg(m.rev, o);
    else q.rev = 0.0;
    q.na = -(V.gam * q.p) * rcp_nr(q.rho);
    return q;
}

__device__ __forceinline__ PcCell pc_dn1(const PcCell& q)
{
    PcCell r;
    r.rho = lane_dn1(q.rho); r.u = lane_dn1(q.u); r.v = lane_dn1(q.v); r.w = lane_dn1(q.w); r.p = lane_dn1(q.p); r.e = lane_dn1(q.e);
    r.na = lane_dn1(q.na); r.rlv = lane_dn1(q.rlv); r.rev = lane_dn1(q.rev);
    return r;
}

// thin-layer viscous flux through the face between L and R (viscousFluxApprox, fluxes.F90:3487-3859): the gradient of a quantity at
// the face is its difference along the centre-to-centre vector dN -- vm_face of kernels_viscous.hip with the nodal gradients zero
__device__ __forceinline__ void pc_tl_face(const PcK& V, const PcCell& L, const PcCell& R, const adf_real8 fN[3], const adf_real8 dN[3],
                                           int por_code, double f[4])
{
    adf_real8 por = V.porV;
    if (por_code == ADF_POR_NOFLUX) por = 0.0;
    const double mul = por * (L.rlv + R.rlv);
    double mue = 0.0;
    if (V.eddy) mue = por * (L.rev + R.rev);
    const double mut = mul + mue;
    const double heatCoef = mul * V.hl + mue * V.ht;
    const adf_real8 ss = rsq_nr(dN[0] * dN[0] + dN[1] * dN[1] + dN[2] * dN[2]);
    const adf_real8 ssx = ss * dN[0], ssy = ss * dN[1], ssz = ss * dN[2];
    const double du = (R.u - L.u) * ss, dv = (R.v - L.v) * ss, dw = (R.w - L.w) * ss, dq = (R.na - L.na) * ss;
    const double u_x = du * ssx, u_y = du * ssy, u_z = du * ssz;
    const double v_x = dv * ssx, v_y = dv * ssy, v_z = dv * ssz;
    const double w_x = dw * ssx, w_y = dw * ssy, w_z = dw * ssz;
    const double q_x = (dq * ssx) * heatCoef, q_y = (dq * ssy) * heatCoef, q_z = (dq * ssz) * heatCoef;
    const double fracDiv = (2.0 * (1.0 / 3.0)) * (u_x + v_y + w_z);
    const double tauxx = mut * (2.0 * u_x - fracDiv), tauyy = mut * (2.0 * v_y - fracDiv), tauzz = mut * (2.0 * w_z - fracDiv);
    const double tauxy = mut * (u_y + v_x), tauxz = mut * (u_z + w_x), tauyz = mut * (v_z + w_y);
    const double ubar = 0.5 * (L.u + R.u), vbar = 0.5 * (L.v + R.v), wbar = 0.5 * (L.w + R.w);
    const adf_real8 nx = fN[0], ny = fN[1], nz = fN[2];
    f[0] = tauxx * nx + tauxy * ny + tauxz * nz;
    f[1] = tauxy * nx + tauyy * ny + tauyz * nz;
    f[2] = tauxz * nx + tauyz * ny + tauzz * nz;
    double frhoE = (ubar * tauxx + vbar * tauxy + wbar * tauxz) * nx;
    frhoE = frhoE + (ubar * tauxy + vbar * tauyy + wbar * tauyz) * ny;
    frhoE = frhoE + (ubar * tauxz + vbar * tauyz + wbar * tauzz) * nz;
    f[3] = frhoE - q_x * nx - q_y * ny - q_z * nz;
}

// G = what leaves L and enters R through the face: central + Roe dissipation - viscous
__device__ __forceinline__ void pc_face(const RmK& K, const PcK& V, const PcCell& L, const PcCell& R, const adf_real8 fN[3],
                                        const adf_real8 dN[3], int por, double G[5])
{
    RCell b, c;
    b.rho = L.rho; b.u = L.u; b.v = L.v; b.w = L.w; b.p = L.p; b.e = L.e;
    c.rho = R.rho; c.u = R.u; c.v = R.v; c.w = R.w; c.p = R.p; c.e = R.e;
    const double Ls[5] = {L.rho, L.u, L.v, L.w, L.p}, Rs[5] = {R.rho, R.u, R.v, R.w, R.p};     // first order: the cell values
    double fc[5], fd[5], fv[4];
    rm_face(K, b, c, Ls, Rs, fN[0], fN[1], fN[2], por, fc, fd);
    pc_tl_face(V, L, R, fN, dN, por, fv);
    G[0] = fc[0] + fd[0];
#pragma unroll
    for (int l = 1; l < 5; ++l) G[l] = (fc[l] + fd[l]) - fv[l - 1];
}

__device__ __forceinline__ void pc_ld3(GPTR(const adf_real8) a, unsigned o, unsigned nb8, adf_real8 v[3])
{
    v[0] = ldg(a, o); v[1] = ldg(a, o + nb8); v[2] = ldg(a, o + 2u * nb8);
}

// SNAP: the Jacobian assembly's snapshot entry instead of dw (KParams::snapTab) -- a compile-time switch: in the dual build only the
// derivative part of the result is stored then, and the value-only arithmetic behind it goes away.
// Every j face is evaluated ONCE: a wave evaluates the j face ABOVE its cell and hands the flux to the row above through LDS; the cell
// is completed one plane later, behind the next barrier, with the flux that arrived through its lower j face (round 5: 3.25 instead of
// the 4 face evaluations of the form with both j faces per cell, which round 6 removed: profiles/r05_x_ab.txt).  The face below row 0 of the tile has no wave: wave (k mod 4) takes it in plane k (row j0-1 loaded by that wave, row
// j0 from the LDS slot of wave 0).  3.25 instead of 4 face evaluations per cell -- what the dual build, bound by FP64 issue, is short of.
template <bool SNAP>
__global__ __launch_bounds__(64 * PM_BY, PM_MINWG) void k_pc_march(const BlkView* __restrict__ tab, const int4* __restrict__ tiles, KParams kp,
                                                            int kch)
{
    __shared__ double qx[2 * PM_BY * PM_NV * 64];       // state of the own cell of every row, by the parity of the plane
    __shared__ double fj[2 * PM_BY * 5 * 64];           // flux handed to every row through its lower j face, by the parity of the plane
    const int4 t = tiles[blockIdx.x];
    if (t.x < 0) return;
    const BlkView& b = tab[t.x];
    const int lane = threadIdx.x, row = threadIdx.y;
    const int i = t.y * PM_OUT + lane;          // columns i0-2 .. i0+61
    const int j = 2 + t.z * PM_BY + row;
    const int k0 = 2 + t.w * kch;
    const int k1 = (k0 + kch - 1 < b.kl) ? k0 + kch - 1 : b.kl;
    const bool out = (lane >= 2 && lane <= 61 && i <= b.il && j <= b.jl);
    const int ic = (i < b.ib) ? i : b.ib, jc = (j < b.je) ? j : b.je;
    const long nb = b.nbox;
    // byte offsets of 8-byte elements, for the geometry and the state alike (the dual forms of ldg / stg double them, kernels_ad.hip)
    unsigned c = 8u * (unsigned)(ic + jc * b.ldi + k0 * b.ldk);
    const unsigned sj = 8u * (unsigned)b.ldi, sk = 8u * (unsigned)b.ldk, nb8 = 8u * (unsigned)nb;
    PcPtrs m;
    m.w0 = (GPTR(const double))b.w; m.w1 = m.w0 + nb; m.w2 = m.w1 + nb; m.w3 = m.w2 + nb; m.w4 = m.w3 + nb;
    m.p = (GPTR(const double))b.p; m.rlv = (GPTR(const double))b.rlv; m.rev = (GPTR(const double))b.rev;
    GPTR(const adf_real8) sI = (GPTR(const adf_real8))b.sI; GPTR(const adf_real8) sJ = (GPTR(const adf_real8))b.sJ;
    GPTR(const adf_real8) sK = (GPTR(const adf_real8))b.sK;
    GPTR(const adf_real8) dI = (GPTR(const adf_real8))b.dI; GPTR(const adf_real8) dJ = (GPTR(const adf_real8))b.dJ;
    GPTR(const adf_real8) dK = (GPTR(const adf_real8))b.dK;
    GPTR(const uint8_t) flags = (GPTR(const uint8_t))b.flags;
    GPTR(double) dw0 = (GPTR(double))b.dw;
    GPTR(double) dw1 = dw0 + nb; GPTR(double) dw2 = dw1 + nb; GPTR(double) dw3 = dw2 + nb; GPTR(double) dw4 = dw3 + nb;

    RmK K;
    K.doDiss = fabs(kp.rFil) >= 1.e-10;
    K.omk = 0.0; K.opk = 0.0; K.factMinmod = 0.0;
    K.gam = kp.gammaConstant; K.gm1 = kp.gammaConstant - 1.0; K.ovgm1 = 1.0 / (kp.gammaConstant - 1.0);
    K.porDiss = 0.5 * kp.rFil;
    PcK V;
    V.porV = 0.5 * kp.rFil; V.eddy = kp.eddyModel != 0; V.gam = kp.gammaConstant;
    V.hl = 1.0 / (kp.prandtl * (kp.gammaConstant - 1.0)); V.ht = 1.0 / (kp.prandtlTurb * (kp.gammaConstant - 1.0));

    PcCell q0 = pc_ld(m, c, V);
    double gk[5];               // what enters the cell through its lower k face
    {
        const PcCell qm1 = pc_ld(m, c - sk, V);
        adf_real8 nK[3], dKv[3];
        pc_ld3(sK, c - sk, nb8, nK); pc_ld3(dK, c - sk, nb8, dKv);
        pc_face(K, V, qm1, q0, nK, dKv, flg_porK(flags[(c - sk) >> 3]), gk);
    }
    // row j0 of the tile in plane k0 (the fifth j face lies below it; not c - row sj: rows beyond the block are clamped)
    const unsigned cE = 8u * (unsigned)(ic + (2 + t.z * PM_BY) * b.ldi + k0 * b.ldk);
    double accP[5] = {0, 0, 0, 0, 0};                   // cell k-1: everything but the flux through its lower j face
    int flagm = 0;
    auto row_state = [&](const double* __restrict__ qb, int r) {
        const double* __restrict__ qi = qb + r * (PM_NV * 64) + lane;
        PcCell q;
        q.rho = qi[0]; q.u = qi[64]; q.v = qi[128]; q.w = qi[192]; q.p = qi[256]; q.e = qi[320]; q.na = qi[384];
        q.rlv = qi[448]; q.rev = qi[512];
        return q;
    };
    // completes the cell of the plane below c with the flux handed over in that plane, and writes it
    auto finish = [&](int kdone) {
        const double* __restrict__ fi = fj + (kdone & 1) * (PM_BY * 5 * 64) + row * (5 * 64) + lane;
        if (out) {
            const unsigned cw = c - sk;
            const adf_real8 blank = (flagm & 64) ? 1.0 : 0.0;
            if (SNAP) {
                const SnapSlot ss = kp.snapTab[t.x];
                const adf_real8 ovol = 1.0 / ldg((GPTR(const adf_real8))b.volRef, cw);
                GPTR(adf_real8) sn = (GPTR(adf_real8))ss.snap + ((long)kp.snapCol * kp.snapN - kp.snapL0) * nb;
#pragma unroll
                for (int l = 0; l < 5; ++l) snap_put(sn + l * nb, cw, ((accP[l] - fi[l * 64]) * blank) * ovol);
            } else {
                stg(dw0, cw, (accP[0] - fi[0]) * blank); stg(dw1, cw, (accP[1] - fi[64]) * blank); stg(dw2, cw, (accP[2] - fi[128]) * blank);
                stg(dw3, cw, (accP[3] - fi[192]) * blank); stg(dw4, cw, (accP[4] - fi[256]) * blank);
            }
        }
    };
    for (int k = k0; k <= k1; ++k) {
        double* __restrict__ qb = qx + (k & 1) * (PM_BY * PM_NV * 64);
        double* __restrict__ fb = fj + (k & 1) * (PM_BY * 5 * 64);
        {
            double* __restrict__ qo = qb + row * (PM_NV * 64) + lane;
            qo[0] = q0.rho; qo[64] = q0.u; qo[128] = q0.v; qo[192] = q0.w; qo[256] = q0.p; qo[320] = q0.e; qo[384] = q0.na;
            qo[448] = q0.rlv; qo[512] = q0.rev;
        }
        const bool fifth = (wave_uniform(row) == (k & 3));      // this wave evaluates the face below row 0 in this plane
        // ---- request: the next plane, the geometry of the three faces, the row above the tile; the fifth face: row j0-1 and its geometry
        const PcCell qp1 = pc_ld(m, c + sk, V);
        const int flag0 = flags[c >> 3];
        // (the cell centres k_visc_gf forms its vectors from -- 3 values per cell where dI / dJ / dK are 9 -- cost this kernel 36 B of
        // scratch per lane, carried or requested every step: it keeps the stored vectors)
        adf_real8 nI[3], dIv[3], nJ[3], dJv[3], nK[3], dKv[3], nE[3], dE[3];
        pc_ld3(sI, c, nb8, nI); pc_ld3(dI, c, nb8, dIv);
        pc_ld3(sJ, c, nb8, nJ); pc_ld3(dJ, c, nb8, dJv);
        pc_ld3(sK, c, nb8, nK); pc_ld3(dK, c, nb8, dKv);
        PcCell qjp, qE;
        int flagE = 0;
        if (row == PM_BY - 1) qjp = pc_ld(m, c + sj, V);
        if (fifth) {
            const unsigned ce = cE + (unsigned)(k - k0) * sk - sj;
            qE = pc_ld(m, ce, V);
            pc_ld3(sJ, ce, nb8, nE); pc_ld3(dJ, ce, nb8, dE);
            flagE = flags[ce >> 3];
        }
        __syncthreads();
        if (k > k0) finish(k - 1);
        double acc[5], G[5];
        // ---- i face (i | i+1); the face (i-1 | i) comes from lane-1
        {
            const PcCell qR = pc_dn1(q0);
            pc_face(K, V, q0, qR, nI, dIv, flg_porI((uint8_t)flag0), G);
#pragma unroll
            for (int l = 0; l < 5; ++l) acc[l] = (G[l] - lane_up1(G[l])) - gk[l];
        }
        // ---- j face (j | j+1): its flux leaves this cell and is handed to the row above
        if (row < PM_BY - 1) qjp = row_state(qb, row + 1);
        pc_face(K, V, q0, qjp, nJ, dJv, flg_porJ((uint8_t)flag0), G);
#pragma unroll
        for (int l = 0; l < 5; ++l) acc[l] += G[l];
        if (row < PM_BY - 1) {
            double* __restrict__ fo = fb + (row + 1) * (5 * 64) + lane;
#pragma unroll
            for (int l = 0; l < 5; ++l) fo[l * 64] = G[l];
        }
        // ---- the fifth j face (j0-1 | j0)
        if (fifth) {
            const PcCell qE0 = row_state(qb, 0);
            pc_face(K, V, qE, qE0, nE, dE, flg_porJ((uint8_t)flagE), G);
            double* __restrict__ fo = fb + lane;
#pragma unroll
            for (int l = 0; l < 5; ++l) fo[l * 64] = G[l];
        }
        // ---- k face above the cell
        pc_face(K, V, q0, qp1, nK, dKv, flg_porK((uint8_t)flag0), gk);
#pragma unroll
        for (int l = 0; l < 5; ++l) accP[l] = acc[l] + gk[l];
        flagm = flag0;
        q0 = qp1;
        c += sk;
    }
    __syncthreads();
    finish(k1);
}


void launch_pc_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, int kch, hipStream_t s)
{
    if (ntiles <= 0) return;
    const dim3 grd(ntiles), blk(64, PM_BY, 1);
    if (kp.snapTab) {
        adf_note_snap(1);
        hipLaunchKernelGGL(k_pc_march<true>, grd, blk, 0, s, tab, tiles, kp, kch);
    } else
        hipLaunchKernelGGL(k_pc_march<false>, grd, blk, 0, s, tab, tiles, kp, kch);
}
