// Viscous residual: Green-Gauss nodal gradients + face stress / heat flux.
//
// Reference semantics:
//   a^2 = gamma p / rho         flowUtils::computeSpeedOfSoundSquared  src/utils/flowUtils.F90:488-550
//   nodal gradients             flowUtils::allNodalGradients           src/utils/flowUtils.F90:1676-2026
//   face flux                   fluxes::viscousFlux                    src/solver/fluxes.F90:2534-3485
//   final sum                   residual_block                         src/solver/residuals.F90:334-344
//
// Both kernels are gathers: a node sums the six integration points of its dual
// cell; a cell evaluates the viscous flux through its six faces (each face is a
// pure function of its surroundings, so the two adjacent cells agree bitwise).
// a^2 is formed on the fly from (gamma, p, rho) instead of being stored.
// Roofline: HBM (255 B/cell RANS); no MFMA.
#include "sa_core.h"

#ifndef ADF_AD_BUILD
extern int g_march_kch;
#endif

#define VS_BX 64
#define VS_BY 4

int g_viscous_tiled = 2;   // tuning "viscous_tiled": 0 = gather kernels, 1 = LDS-tiled nodal-gradient and face-flux kernels, 2 = k-marching nodal gradients

__device__ __forceinline__ double aa_at(const BlkView& b, long q) { return b.gamma[q] * b.p[q] / b.w[q]; }

struct NCell { double u, v, w, aa; };

// The two integration points of one direction of the dual-cell surface integral
// (flowUtils.F90:1712-1791 for k).  The point below the node (sign -) uses the 2x2
// patch of cells m[0..3] and the face normals at index-1 and index of those cells,
// the point above it (sign +) the patch p[0..3] and the normals at index and
// index+1: the normals at `index` are shared, every cell value is loaded once by
// the caller.  Summation order as in the reference: the four faces at the lower
// index first, then the four at the upper one.
__device__ __forceinline__ void grad_dir(const BlkView& b, long c, long sd, long s1, long s2, const adf_real8* __restrict__ sN,
                                         const NCell& m0, const NCell& m1, const NCell& m2, const NCell& m3, const NCell& p0,
                                         const NCell& p1, const NCell& p2, const NCell& p3, double g[12])
{
    const long nb = b.nbox;
    const long cc[4] = {c, c + s1, c + s2, c + s1 + s2};
    double lo[3] = {0.0, 0.0, 0.0}, mid[4][3], hi[4][3];
    double sm[3] = {0.0, 0.0, 0.0}, sp[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            sm[d] += sN[cc[q] - sd + d * nb];
            mid[q][d] = sN[cc[q] + d * nb];
            hi[q][d] = sN[cc[q] + sd + d * nb];
        }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int d = 0; d < 3; ++d) { sm[d] += mid[q][d]; sp[d] += mid[q][d]; }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int d = 0; d < 3; ++d) sp[d] += hi[q][d];
    (void)lo;
    {
        const double ubar = 0.25 * (m0.u + m1.u + m2.u + m3.u), vbar = 0.25 * (m0.v + m1.v + m2.v + m3.v);
        const double wbar = 0.25 * (m0.w + m1.w + m2.w + m3.w), a2 = 0.25 * (m0.aa + m1.aa + m2.aa + m3.aa);
        g[0] += -1.0 * ubar * sm[0]; g[1] += -1.0 * ubar * sm[1]; g[2] += -1.0 * ubar * sm[2];
        g[3] += -1.0 * vbar * sm[0]; g[4] += -1.0 * vbar * sm[1]; g[5] += -1.0 * vbar * sm[2];
        g[6] += -1.0 * wbar * sm[0]; g[7] += -1.0 * wbar * sm[1]; g[8] += -1.0 * wbar * sm[2];
        g[9] -= -1.0 * a2 * sm[0]; g[10] -= -1.0 * a2 * sm[1]; g[11] -= -1.0 * a2 * sm[2];
    }
    {
        const double ubar = 0.25 * (p0.u + p1.u + p2.u + p3.u), vbar = 0.25 * (p0.v + p1.v + p2.v + p3.v);
        const double wbar = 0.25 * (p0.w + p1.w + p2.w + p3.w), a2 = 0.25 * (p0.aa + p1.aa + p2.aa + p3.aa);
        g[0] += ubar * sp[0]; g[1] += ubar * sp[1]; g[2] += ubar * sp[2];
        g[3] += vbar * sp[0]; g[4] += vbar * sp[1]; g[5] += vbar * sp[2];
        g[6] += wbar * sp[0]; g[7] += wbar * sp[1]; g[8] += wbar * sp[2];
        g[9] -= a2 * sp[0]; g[10] -= a2 * sp[1]; g[11] -= a2 * sp[2];
    }
}

// gradient of the node stored at cell index c (nodes 1..il x 1..jl x 1..kl; node (i,j,k) is stored at cell index (i,j,k))
__device__ __forceinline__ void node_gradient(const BlkView& b, long c)
{
    const long nb = b.nbox;
    const long si = 1, sj = b.ldi, sk = b.ldk;
    // the eight cells around the node, index di + 2 dj + 4 dk
    NCell q[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const long cq = c + (n & 1) * si + ((n >> 1) & 1) * sj + (n >> 2) * sk;
        q[n].u = b.w[cq + nb]; q[n].v = b.w[cq + 2 * nb]; q[n].w = b.w[cq + 3 * nb];
        q[n].aa = aa_at(b, cq);
    }
    double g[12];
#pragma unroll
    for (int m = 0; m < 12; ++m) g[m] = 0.0;
    // k-direction (flowUtils.F90:1759-1791): patches (i..i+1, j..j+1) at k and k+1
    grad_dir(b, c, sk, si, sj, b.sK, q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], g);
    // j-direction (:1805-1892): patches (i..i+1, k..k+1) at j and j+1
    grad_dir(b, c, sj, si, sk, b.sJ, q[0], q[1], q[4], q[5], q[2], q[3], q[6], q[7], g);
    // i-direction (:1894-1979): patches (j..j+1, k..k+1) at i and i+1
    grad_dir(b, c, si, sj, sk, b.sI, q[0], q[2], q[4], q[6], q[1], q[3], q[5], q[7], g);
    const double oneOverV = 1.0 / (b.vol[c] + b.vol[c + sk] + b.vol[c + si] + b.vol[c + si + sk] + b.vol[c + sj] +
                                   b.vol[c + sj + sk] + b.vol[c + si + sj] + b.vol[c + si + sj + sk]);
#pragma unroll
    for (int m = 0; m < 12; ++m) b.grad[c + m * nb] = g[m] * oneOverV;
}

__global__ __launch_bounds__(VS_BX* VS_BY) void k_nodal_gradients(BlkView b)
{
    const int i = blockIdx.x * VS_BX + threadIdx.x + (2 - 16);   // aligned rows, see internal.h
    const int j = blockIdx.y * VS_BY + threadIdx.y + 1;
    const int k = blockIdx.z + 1;
    if (i < 1 || i > b.il || j > b.jl) return;
    node_gradient(b, b.idx(i, j, k));
}

// (the marching forms below compile in the forward-mode build too: round 6, k_visc_gf on dual numbers)
// ---------------------------------------------------------------------------
// The dual-cell surface integral of allNodalGradients (flowUtils.F90:1712-1979) factorises: with the per-CELL vectors
// tI = sI(i-1) + sI(i), tJ = sJ(j-1) + sJ(j), tK = sK(k-1) + sK(k) the normal of the integration point of direction k at cell plane m is
// the sum of tK over the 2 x 2 cells around the node column, the one of direction j at cell row m the sum of tJ over
// (i..i+1) x (k..k+1), and likewise for i; the averaged state is the sum of (u, v, w, -a^2) over the same four cells.  A thread
// marching in k needs one cell record per plane and row (k_visc_gf below); the sums are formed in another order than the
// reference's: results agree to rounding.
// ---------------------------------------------------------------------------
#define NG_BY 4
#define NG_KCH 32

// g += (NEG ? -1 : +1) phi (x) t   (the factor 0.25 of the surface integral rides on 1 / volume in the caller)
template <bool NEG>
__device__ __forceinline__ void ng_outer(double g[12], const double ph[4], const adf_real8 t[3])
{
#pragma unroll
    for (int v = 0; v < 4; ++v) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (NEG) g[3 * v + d] -= ph[v] * t[d];
            else g[3 * v + d] += ph[v] * t[d];
        }
    }
}

__device__ __forceinline__ void vm_ld3(GPTR(const adf_real8) a, unsigned o, unsigned nb8, adf_real8 v[3])
{
    v[0] = ldg(a, o); v[1] = ldg(a, o + nb8); v[2] = ldg(a, o + 2 * nb8);
}


// viscous flux through the face between cells cL and cL+sd (fluxes.F90:2610-2860);
// the four face nodes are cL, cL-s1, cL-s2, cL-s1-s2.
__device__ __forceinline__ void visc_face(const BlkView& b, const KParams& kp, long cL, long sd, long s1, long s2,
                                          const adf_real8* __restrict__ sN, int por_code, double sign, double acc[5],
                                          double* tq = nullptr)
{
    const long nb = b.nbox;
    const long cR = cL + sd;
    double por = 0.5 * kp.rFil;
    if (por_code == ADF_POR_NOFLUX) por = 0.0;
    const double mul = por * (b.rlv[cL] + b.rlv[cR]);
    double mue = 0.0;
    if (kp.eddyModel) mue = por * (b.rev[cL] + b.rev[cR]);
    const double mut = mul + mue;
    const double gm1 = 0.5 * (b.gamma[cL] + b.gamma[cR]) - 1.0;
    const double heatCoef = mul * (1.0 / (kp.prandtl * gm1)) + mue * (1.0 / (kp.prandtlTurb * gm1));

    // reference order of the four nodes: (-s1-s2), (-s2), (-s1), (0)
    const long n0 = cL - s1 - s2, n1 = cL - s2, n2 = cL - s1, n3 = cL;
    double gr[12];
#pragma unroll
    for (int m = 0; m < 12; ++m) {
        const double* g = b.grad + m * nb;
        gr[m] = 0.25 * (g[n0] + g[n1] + g[n2] + g[n3]);
    }
    // vector between the two cell centres (reference node order kept)
    double ssv[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const adf_real8* xx = b.x + d * nb;
        ssv[d] = 0.125 * (xx[n0 + sd] - xx[n0 - sd] + xx[n2 + sd] - xx[n2 - sd] + xx[n1 + sd] - xx[n1 - sd] + xx[n3 + sd] -
                          xx[n3 - sd]);
    }
    const double ss = 1.0 / sqrt(ssv[0] * ssv[0] + ssv[1] * ssv[1] + ssv[2] * ssv[2]);
    const double ssx = ss * ssv[0], ssy = ss * ssv[1], ssz = ss * ssv[2];
    const double uL = b.w[cL + nb], vL = b.w[cL + 2 * nb], wL = b.w[cL + 3 * nb];
    const double uR = b.w[cR + nb], vR = b.w[cR + 2 * nb], wR = b.w[cR + 3 * nb];
    double corr;
    corr = gr[0] * ssx + gr[1] * ssy + gr[2] * ssz - (uR - uL) * ss;
    double u_x = gr[0] - corr * ssx, u_y = gr[1] - corr * ssy, u_z = gr[2] - corr * ssz;
    corr = gr[3] * ssx + gr[4] * ssy + gr[5] * ssz - (vR - vL) * ss;
    double v_x = gr[3] - corr * ssx, v_y = gr[4] - corr * ssy, v_z = gr[5] - corr * ssz;
    corr = gr[6] * ssx + gr[7] * ssy + gr[8] * ssz - (wR - wL) * ss;
    double w_x = gr[6] - corr * ssx, w_y = gr[7] - corr * ssy, w_z = gr[8] - corr * ssz;
    corr = gr[9] * ssx + gr[10] * ssy + gr[11] * ssz + (aa_at(b, cR) - aa_at(b, cL)) * ss;
    double q_x = gr[9] - corr * ssx, q_y = gr[10] - corr * ssy, q_z = gr[11] - corr * ssz;

    const double fracDiv = (2.0 * (1.0 / 3.0)) * (u_x + v_y + w_z);
    const double tauxxS = 2.0 * u_x - fracDiv, tauyyS = 2.0 * v_y - fracDiv, tauzzS = 2.0 * w_z - fracDiv;
    const double tauxyS = u_y + v_x, tauxzS = u_z + w_x, tauyzS = v_z + w_y;
    q_x *= heatCoef; q_y *= heatCoef; q_z *= heatCoef;
    double tauxx, tauyy, tauzz, tauxy, tauxz, tauyz;
    if (kp.useQCR) {
        double den = sqrt(u_x * u_x + u_y * u_y + u_z * u_z + v_x * v_x + v_y * v_y + v_z * v_z + w_x * w_x + w_y * w_y +
                          w_z * w_z);
        den = fmax(den, 1.e-14);
        const double fact = mue * 0.3 / den;
        const double Wxy = u_y - v_x, Wxz = u_z - w_x, Wyz = v_z - w_y;
        const double Wyx = -Wxy, Wzx = -Wxz, Wzy = -Wyz;
        const double exx = fact * (Wxy * tauxyS + Wxz * tauxzS) * 2.0;
        const double eyy = fact * (Wyx * tauxyS + Wyz * tauyzS) * 2.0;
        const double ezz = fact * (Wzx * tauxzS + Wzy * tauyzS) * 2.0;
        const double exy = fact * (Wxy * tauyyS + Wxz * tauyzS + Wyx * tauxxS + Wyz * tauxzS);
        const double exz = fact * (Wxy * tauyzS + Wxz * tauzzS + Wzx * tauxxS + Wzy * tauxyS);
        const double eyz = fact * (Wyx * tauxzS + Wyz * tauzzS + Wzx * tauxyS + Wzy * tauyyS);
        tauxx = mut * tauxxS - exx; tauyy = mut * tauyyS - eyy; tauzz = mut * tauzzS - ezz;
        tauxy = mut * tauxyS - exy; tauxz = mut * tauxzS - exz; tauyz = mut * tauyzS - eyz;
    } else {
        tauxx = mut * tauxxS; tauyy = mut * tauyyS; tauzz = mut * tauzzS;
        tauxy = mut * tauxyS; tauxz = mut * tauxzS; tauyz = mut * tauyzS;
    }
    if (tq) {   // wall stress tensor / heat flux of a viscous subface (fluxes.F90:2861-2892)
        tq[0] = tauxx; tq[1] = tauyy; tq[2] = tauzz; tq[3] = tauxy; tq[4] = tauxz; tq[5] = tauyz;
        tq[6] = q_x; tq[7] = q_y; tq[8] = q_z;
        return;
    }
    const double ubar = 0.5 * (uL + uR), vbar = 0.5 * (vL + vR), wbar = 0.5 * (wL + wR);
    const double nx = sN[cL], ny = sN[cL + nb], nz = sN[cL + 2 * nb];
    const double fmx = tauxx * nx + tauxy * ny + tauxz * nz;
    const double fmy = tauxy * nx + tauyy * ny + tauyz * nz;
    const double fmz = tauxz * nx + tauyz * ny + tauzz * nz;
    double frhoE = (ubar * tauxx + vbar * tauxy + wbar * tauxz) * nx;
    frhoE = frhoE + (ubar * tauxy + vbar * tauyy + wbar * tauyz) * ny;
    frhoE = frhoE + (ubar * tauxz + vbar * tauyz + wbar * tauzz) * nz;
    frhoE = frhoE - q_x * nx - q_y * ny - q_z * nz;
    // fw(left) -= f ; fw(right) += f
    acc[1] += sign * fmx;
    acc[2] += sign * fmy;
    acc[3] += sign * fmz;
    acc[4] += sign * frhoE;
}

// owned cells: fw += viscous fluxes; dw = (dw + fw) * iblank
__global__ __launch_bounds__(VS_BX* VS_BY) void k_viscous(BlkView b, KParams kp)
{
    const int i = blockIdx.x * VS_BX + threadIdx.x + 2;
    const int j = blockIdx.y * VS_BY + threadIdx.y + 2;
    const int k = blockIdx.z + 2;
    if (i > b.il || j > b.jl) return;
    const long c = b.idx(i, j, k);
    const long nb = b.nbox;
    const long si = 1, sj = b.ldi, sk = b.ldk;
    const uint8_t f0 = b.flags[c];
    const uint8_t fi = b.flags[c - si], fj = b.flags[c - sj], fk = b.flags[c - sk];
    double acc[5] = {0, 0, 0, 0, 0};
    // reference sweep order k, j, i (fluxes.F90:2610, 2903, 3197).  The three
    // directions run as a rolled loop: unrolled, the six inlined face evaluations
    // need > 256 VGPRs and spill.
    const long sd3[3] = {sk, sj, si};
    const long s13[3] = {si, si, sj};
    const long s23[3] = {sj, sk, sk};
    const adf_real8* sN3[3] = {b.sK, b.sJ, b.sI};
    const int porM3[3] = {flg_porK(fk), flg_porJ(fj), flg_porI(fi)};
    const int porP3[3] = {flg_porK(f0), flg_porJ(f0), flg_porI(f0)};
#pragma unroll 1
    for (int d = 0; d < 3; ++d) {
        visc_face(b, kp, c - sd3[d], sd3[d], s13[d], s23[d], sN3[d], porM3[d], +1.0, acc);
        visc_face(b, kp, c, sd3[d], s13[d], s23[d], sN3[d], porP3[d], -1.0, acc);
    }
    const double blank = flg_blank(f0);
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        const double fwn = (kp.fwMode ? b.fw[c + l * nb] : 0.0) + acc[l];   // without fwMode dw already holds dw + fw
        if (kp.fwMode) b.fw[c + l * nb] = fwn;
        b.dw[c + l * nb] = (b.dw[c + l * nb] + fwn) * blank;
    }
}

#ifndef ADF_AD_BUILD
// Nodal gradients of the node plane ON a viscous-wall subface: the only gradients the stored wall stress reads (the four nodes of a
// wall face, fluxes.F90:2849-2879).  k_visc_gf keeps its gradients in LDS; instead of writing all twelve of every node of the block
// for the sake of the wall faces (96 B per cell), the few nodes of the wall planes are formed again here, in the reference's own
// summation order (k_nodal_gradients).  blockIdx.y = subface entry of the level's boundary plan.
__global__ __launch_bounds__(256) void k_wall_node_grad(const BlkView* __restrict__ tab, const BcEntry* __restrict__ ent,
                                                        const int* __restrict__ order)
{
    const BcEntry& e = ent[order[blockIdx.y]];
    const BcFaceDev& f = e.f;
    if (!f.tauq) return;
    const BlkView& b = tab[e.slot];
    int r[4];
    bc_owned_range(f.faceID, f.icBeg, f.icEnd, f.jcBeg, f.jcEnd, b.il, b.jl, b.kl, r);
    const int na = r[1] - r[0] + 2, nbb = r[3] - r[2] + 2;         // nodes r[0]-1 .. r[1]
    if (na <= 1 || nbb <= 1) return;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)na * nbb) return;
    const int a = r[0] - 1 + (int)(t % na), bb = r[2] - 1 + (int)(t / na);
    long c;
    switch (f.faceID) {
    case ADFLOW_IMIN: case ADFLOW_IMAX: c = b.idx(f.faceID == ADFLOW_IMIN ? 1 : b.il, a, bb); break;
    case ADFLOW_JMIN: case ADFLOW_JMAX: c = b.idx(a, f.faceID == ADFLOW_JMIN ? 1 : b.jl, bb); break;
    default: c = b.idx(a, bb, f.faceID == ADFLOW_KMIN ? 1 : b.kl);
    }
    node_gradient(b, c);
}

// Stress tensor and heat flux vector on the faces of the viscous-wall subfaces: what viscousFlux stores in
// viscSubface(:)%tau / %q when storeWallTensor is set (rkStage == 0 on the ground level, fluxes.F90:2586-2592,
// 2861-2892, 3155-3185, 3450-3480).  Same face evaluation as the flux kernels, from the nodal gradients they left in
// b.grad; blockIdx.y = subface entry of the level's boundary plan.
__global__ __launch_bounds__(256) void k_wall_stress(const BlkView* __restrict__ tab, const BcEntry* __restrict__ ent,
                                                     const int* __restrict__ order, KParams kp)
{
    const BcEntry& e = ent[order[blockIdx.y]];
    const BcFaceDev& f = e.f;
    if (!f.tauq) return;
    const BlkView& b = tab[e.slot];
    int r[4];
    bc_owned_range(f.faceID, f.icBeg, f.icEnd, f.jcBeg, f.jcEnd, b.il, b.jl, b.kl, r);
    const int na = r[1] - r[0] + 1, nbb = r[3] - r[2] + 1;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (na <= 0 || nbb <= 0 || t >= (long)na * nbb) return;
    const int a = r[0] + (int)(t % na), bb = r[2] + (int)(t / na);
    const long si = 1, sj = b.ldi, sk = b.ldk;
    long cL, sd, s1, s2;
    const adf_real8* sN;
    int por;
    switch (f.faceID) {
    case ADFLOW_IMIN: case ADFLOW_IMAX:
        cL = b.idx(f.faceID == ADFLOW_IMIN ? 1 : b.il, a, bb); sd = si; s1 = sj; s2 = sk; sN = b.sI; por = flg_porI(b.flags[cL]);
        break;
    case ADFLOW_JMIN: case ADFLOW_JMAX:
        cL = b.idx(a, f.faceID == ADFLOW_JMIN ? 1 : b.jl, bb); sd = sj; s1 = si; s2 = sk; sN = b.sJ; por = flg_porJ(b.flags[cL]);
        break;
    default:
        cL = b.idx(a, bb, f.faceID == ADFLOW_KMIN ? 1 : b.kl); sd = sk; s1 = si; s2 = sj; sN = b.sK; por = flg_porK(b.flags[cL]);
    }
    double tq[9], acc[5];
    visc_face(b, kp, cL, sd, s1, s2, sN, por, 1.0, acc, tq);
    const long n = (long)na * nbb;
#pragma unroll
    for (int m = 0; m < 9; ++m) f.tauq[m * n + t] = tq[m];
}

void launch_wall_stress(const BlkView* tab, const BcEntry* ent, const int* order, const BcPhase& ph, const KParams& kp, bool formGrad,
                        hipStream_t s)
{
    if (ph.count <= 0 || ph.maxCells <= 0) return;
    // formGrad: the flux kernel kept its gradients on chip -> the node planes of the wall faces are formed here.  maxCells counts the
    // cells of a subface incl. its halo ring (>= the (n1 + 1) (n2 + 1) nodes of its owned faces)
    if (formGrad)
        hipLaunchKernelGGL(k_wall_node_grad, dim3((unsigned)((ph.maxCells + 255) / 256), ph.count, 1), dim3(256, 1, 1), 0, s, tab, ent,
                           order + ph.first);
    hipLaunchKernelGGL(k_wall_stress, dim3((unsigned)((ph.maxCells + 255) / 256), ph.count, 1), dim3(256, 1, 1), 0, s, tab, ent,
                       order + ph.first, kp);
}
#endif

// ---------------------------------------------------------------------------
// Tiled form of the face-flux kernel.  The gather form above issues ~560 loads
// per cell (48 nodal gradients + 24 node coordinates per face, six faces): the
// texture-address unit, not HBM, bounds it.  Here
//   * the vector between the two cell centres of every face (the only use of
//     the node coordinates x, fluxes.F90:2660-2676) is static geometry, formed
//     once per mesh by k_face_vectors with the reference's summation order;
//   * the nodal gradients of the two node planes a workgroup's 64x4 cells touch
//     are staged ONCE through LDS by coalesced row loads (120 rows of 65 nodes)
//     and the 4-node averages of the six faces read them from there;
//   * the state of a neighbour cell is loaded once per direction, not per face.
// ~125 global loads per cell remain.  Arithmetic (and its order) is unchanged.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(VS_BX* VS_BY) void k_face_vectors(BlkView b)
{
    const int i = blockIdx.x * VS_BX + threadIdx.x + 1;
    const int j = blockIdx.y * VS_BY + threadIdx.y + 1;
    const int k = blockIdx.z + 1;
    if (i > b.ie || j > b.je) return;
    const long c = b.idx(i, j, k);
    const long nb = b.nbox;
    const long si = 1, sj = b.ldi, sk = b.ldk;
    // cell centre: the mean of the eight corner nodes.  k_visc_gf takes the vector between the centres of two neighbouring cells as
    // the difference of two of these (3 values per cell instead of the 9 of dI / dJ / dK); against the eight-term sum of
    // fluxes.F90:2673-2690 the difference carries the rounding of |x| a few times more -- the same order as that sum's own
    {
        const long n0 = c - si - sj - sk;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const adf_real8* xx = b.x + m * nb;
            b.xc[c + m * nb] = 0.125 * (((xx[n0] + xx[n0 + si]) + (xx[n0 + sj] + xx[n0 + si + sj])) +
                                        ((xx[n0 + sk] + xx[n0 + si + sk]) + (xx[n0 + sj + sk] + xx[c])));
        }
    }
    if (i > b.il || j > b.jl || k > b.kl) return;
    const long sd3[3] = {si, sj, sk};
    const long s13[3] = {sj, si, si};
    const long s23[3] = {sk, sk, sj};
    adf_real8* out3[3] = {b.dI, b.dJ, b.dK};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const long sd = sd3[d], s1 = s13[d], s2 = s23[d];
        const long n0 = c - s1 - s2, n1 = c - s2, n2 = c - s1, n3 = c;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const adf_real8* xx = b.x + m * nb;
            out3[d][c + m * nb] = 0.125 * (xx[n0 + sd] - xx[n0 - sd] + xx[n2 + sd] - xx[n2 - sd] + xx[n1 + sd] - xx[n1 - sd] +
                                           xx[n3 + sd] - xx[n3 - sd]);
        }
    }
}

// derived static geometry of the viscous path: the face vectors dI / dJ / dK and the cell centres
void launch_face_vectors(const BlkView& b, hipStream_t s)
{
    dim3 blk(VS_BX, VS_BY, 1);
    dim3 g((b.ie + VS_BX - 1) / VS_BX, (b.je + VS_BY - 1) / VS_BY, b.ke);
    hipLaunchKernelGGL(k_face_vectors, g, blk, 0, s, b);
}
struct VCell { double u, v, w, aa, rlv, rev, gam; };

__device__ __forceinline__ VCell vcell_at(const BlkView& b, const KParams& kp, long q)
{
    const long nb = b.nbox;
    VCell c;
    c.u = b.w[q + nb]; c.v = b.w[q + 2 * nb]; c.w = b.w[q + 3 * nb];
    c.gam = b.gamma[q];
    c.aa = c.gam * b.p[q] / b.w[q];
    c.rlv = b.rlv[q];
    c.rev = kp.eddyModel ? b.rev[q] : 0.0;
    return c;
}

// ---------------------------------------------------------------------------
// Face arithmetic of the marching viscous kernels (k_visc_gf, k_visc_approx_march): vm_face = fluxes.F90:2610-2860 with the SUM of
// the gradients of the four face nodes handed in (the 4-node average is formed from pair sums: rounding only), the constant gamma
// of the path in the heat-conduction factors and 1/|d| from v_rsq_f64.
// ---------------------------------------------------------------------------
#define VM_OUT 60          // tile table shared with the inviscid marching kernels
#define VM_BY 4

struct VmCell { double u, v, w, na, rlv, rev; };      // na = - gamma p / rho (minus the speed of sound squared)

struct VmPtrs {
    GPTR(const double) w0; GPTR(const double) w1; GPTR(const double) w2; GPTR(const double) w3; GPTR(const double) p;
    GPTR(const double) rlv; GPTR(const double) rev;
};

__device__ __forceinline__ VmCell vm_ld(const VmPtrs& m, unsigned o, double gam, bool eddy)
{
    VmCell q;
    q.u = ldg(m.w1, o); q.v = ldg(m.w2, o); q.w = ldg(m.w3, o);
    q.na = -(gam * ldg(m.p, o)) * rcp_nr(ldg(m.w0, o));
    q.rlv = ldg(m.rlv, o);
    q.rev = eddy ? ldg(m.rev, o) : 0.0;
    return q;
}

__device__ __forceinline__ VmCell vm_dn1(const VmCell& q)
{
    VmCell r;
    r.u = lane_dn1(q.u); r.v = lane_dn1(q.v); r.w = lane_dn1(q.w); r.na = lane_dn1(q.na); r.rlv = lane_dn1(q.rlv); r.rev = lane_dn1(q.rev);
    return r;
}

struct VmK { adf_real8 porV, hl, ht; bool eddy; };       // 0.5 rFil; 1 / (prandtl (gamma-1)); 1 / (prandtlTurb (gamma-1))

// viscous flux through the face between L and R (normal fN pointing from L to R, centre-to-centre vector dN); gr: AVERAGE of the
// gradients of the four face nodes (k_visc_gf keeps a QUARTER of every nodal gradient in its ring -- the factor rides on 1 / volume --
// so the average is the plain sum of four ring entries: exact, powers of two)
template <bool QCR>
__device__ __forceinline__ void vm_face(const VmK& K, const double gr[12], const VmCell& L, const VmCell& R, const adf_real8 fN[3],
                                        const adf_real8 dN[3], int por_code, double f[4])
{
    adf_real8 por = K.porV;
    if (por_code == ADF_POR_NOFLUX) por = 0.0;
    const double mul = por * (L.rlv + R.rlv);
    const double mue = K.eddy ? por * (L.rev + R.rev) : 0.0;
    const double mut = mul + mue;
    const double heatCoef = mul * K.hl + mue * K.ht;
    const adf_real8 ss = rsq_nr(dN[0] * dN[0] + dN[1] * dN[1] + dN[2] * dN[2]);
    const adf_real8 ssx = ss * dN[0], ssy = ss * dN[1], ssz = ss * dN[2];
    double corr;
    corr = gr[0] * ssx + gr[1] * ssy + gr[2] * ssz - (R.u - L.u) * ss;
    const double u_x = gr[0] - corr * ssx, u_y = gr[1] - corr * ssy, u_z = gr[2] - corr * ssz;
    corr = gr[3] * ssx + gr[4] * ssy + gr[5] * ssz - (R.v - L.v) * ss;
    const double v_x = gr[3] - corr * ssx, v_y = gr[4] - corr * ssy, v_z = gr[5] - corr * ssz;
    corr = gr[6] * ssx + gr[7] * ssy + gr[8] * ssz - (R.w - L.w) * ss;
    const double w_x = gr[6] - corr * ssx, w_y = gr[7] - corr * ssy, w_z = gr[8] - corr * ssz;
    corr = gr[9] * ssx + gr[10] * ssy + gr[11] * ssz - (R.na - L.na) * ss;
    double q_x = gr[9] - corr * ssx, q_y = gr[10] - corr * ssy, q_z = gr[11] - corr * ssz;
    const double fracDiv = (2.0 * (1.0 / 3.0)) * (u_x + v_y + w_z);
    const double tauxxS = 2.0 * u_x - fracDiv, tauyyS = 2.0 * v_y - fracDiv, tauzzS = 2.0 * w_z - fracDiv;
    const double tauxyS = u_y + v_x, tauxzS = u_z + w_x, tauyzS = v_z + w_y;
    q_x *= heatCoef; q_y *= heatCoef; q_z *= heatCoef;
    double tauxx = mut * tauxxS, tauyy = mut * tauyyS, tauzz = mut * tauzzS;
    double tauxy = mut * tauxyS, tauxz = mut * tauxzS, tauyz = mut * tauyzS;
    if (QCR) {
        double den = sqrt(u_x * u_x + u_y * u_y + u_z * u_z + v_x * v_x + v_y * v_y + v_z * v_z + w_x * w_x + w_y * w_y + w_z * w_z);
        den = fmax(den, 1.e-14);
        const double fact = mue * 0.3 / den;
        const double Wxy = u_y - v_x, Wxz = u_z - w_x, Wyz = v_z - w_y;
        const double Wyx = -Wxy, Wzx = -Wxz, Wzy = -Wyz;
        tauxx -= fact * (Wxy * tauxyS + Wxz * tauxzS) * 2.0;
        tauyy -= fact * (Wyx * tauxyS + Wyz * tauyzS) * 2.0;
        tauzz -= fact * (Wzx * tauxzS + Wzy * tauyzS) * 2.0;
        tauxy -= fact * (Wxy * tauyyS + Wxz * tauyzS + Wyx * tauxxS + Wyz * tauxzS);
        tauxz -= fact * (Wxy * tauyzS + Wxz * tauzzS + Wzx * tauxxS + Wzy * tauxyS);
        tauyz -= fact * (Wyx * tauxzS + Wyz * tauzzS + Wzx * tauxyS + Wzy * tauyyS);
    }
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

#ifdef ADF_AD_BUILD
#define VMA_MINWG 1
#else
#define VMA_MINWG 2
#endif
// (compiles for dual numbers too -- the preconditioner matrix of the scalar / matrix schemes by forward mode: state, LDS rows and
//  fluxes dual, normals and centre-to-centre vectors plain)
// viscousFluxApprox (fluxes.F90:3487-3859), the thin-layer form of the preconditioner assembly, as a k-march over the level's tile
// table: the face gradient is the difference of the two cell values along the centre-to-centre vector, i.e. vm_face with the nodal
// gradients set to zero -- no gradients, no LDS ring.  k faces carried, i faces once (DPP hand-over of the flux), both j faces per
// cell; the state of the j neighbours through LDS.
// FIRST: the kernel runs before the inviscid march: its flux sums are stored to dw(2:5) as they are, the inviscid march (ADDV) adds
// them to its own sums and applies iblank
template <bool FIRST>
__global__ __launch_bounds__(64 * VM_BY, VMA_MINWG) void k_visc_approx_march(const BlkView* __restrict__ tab, const int4* __restrict__ tiles,
                                                                     KParams kp, int kch)
{
    __shared__ double qx[VM_BY * 6 * 64];               // state of the own cell of every row, for the rows above and below
    const int4 t = tiles[blockIdx.x];
    if (t.x < 0) return;
    const BlkView& b = tab[t.x];
    const int lane = threadIdx.x, row = threadIdx.y;
    const int i = t.y * VM_OUT + lane;          // columns i0-2 .. i0+61
    const int j = 2 + t.z * VM_BY + row;
    const int k0 = 2 + t.w * kch;
    const int k1 = (k0 + kch - 1 < b.kl) ? k0 + kch - 1 : b.kl;
    const bool out = (lane >= 2 && lane <= 61 && i <= b.il && j <= b.jl);
    const int ic = (i < b.ib) ? i : b.ib, jc = (j < b.je) ? j : b.je;
    const long nb = b.nbox;
    const unsigned nb8 = 8u * (unsigned)nb, sj = 8u * (unsigned)b.ldi, sk = 8u * (unsigned)b.ldk;
    unsigned c = 8u * (unsigned)(ic + jc * b.ldi + k0 * b.ldk);
    VmPtrs m;
    m.w0 = (GPTR(const double))b.w; m.w1 = m.w0 + nb; m.w2 = m.w1 + nb; m.w3 = m.w2 + nb;
    m.p = (GPTR(const double))b.p; m.rlv = (GPTR(const double))b.rlv; m.rev = (GPTR(const double))b.rev;
    GPTR(const adf_real8) sI = (GPTR(const adf_real8))b.sI; GPTR(const adf_real8) sJ = (GPTR(const adf_real8))b.sJ;
    GPTR(const adf_real8) sK = (GPTR(const adf_real8))b.sK;
    GPTR(const adf_real8) dI = (GPTR(const adf_real8))b.dI; GPTR(const adf_real8) dJ = (GPTR(const adf_real8))b.dJ;
    GPTR(const adf_real8) dK = (GPTR(const adf_real8))b.dK;
    GPTR(const uint8_t) flags = (GPTR(const uint8_t))b.flags;
    GPTR(double) dw = (GPTR(double))b.dw;
    GPTR(double) fw = (GPTR(double))b.fw;
    VmK K;
    K.porV = 0.5 * kp.rFil; K.eddy = kp.eddyModel != 0;
    K.hl = 1.0 / (kp.prandtl * (kp.gammaConstant - 1.0)); K.ht = 1.0 / (kp.prandtlTurb * (kp.gammaConstant - 1.0));
    const adf_real8 gam = kp.gammaConstant;
    double gs[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) gs[q] = 0.0;

    const VmCell qm1 = vm_ld(m, c - sk, gam, K.eddy);
    VmCell q0 = vm_ld(m, c, gam, K.eddy);
    double fk[4];
    {
        // k face below the first plane of the march
        adf_real8 nK[3], dKv[3];
        vm_ld3(sK, c - sk, nb8, nK); vm_ld3(dK, c - sk, nb8, dKv);
        vm_face<false>(K, gs, qm1, q0, nK, dKv, flg_porK(flags[(c - sk) >> 3]), fk);
    }
    for (int k = k0; k <= k1; ++k) {
        {
            double* __restrict__ qo = qx + row * (6 * 64) + lane;
            qo[0] = q0.u; qo[64] = q0.v; qo[128] = q0.w; qo[192] = q0.na; qo[256] = q0.rlv; qo[320] = q0.rev;
        }
        const VmCell qp1 = vm_ld(m, c + sk, gam, K.eddy);
        const int flag0 = flags[c >> 3];
        double acc[4];
        __syncthreads();
        auto row_state = [&](int r) {
            const double* __restrict__ qi = qx + r * (6 * 64) + lane;
            VmCell q;
            q.u = qi[0]; q.v = qi[64]; q.w = qi[128]; q.na = qi[192]; q.rlv = qi[256]; q.rev = qi[320];
            return q;
        };
        // ---- j face (j-1 | j)
        {
            adf_real8 nJ[3], dJv[3]; double f[4];
            const VmCell qjm = (row > 0) ? row_state(row - 1) : vm_ld(m, c - sj, gam, K.eddy);
            vm_ld3(sJ, c - sj, nb8, nJ); vm_ld3(dJ, c - sj, nb8, dJv);
            vm_face<false>(K, gs, qjm, q0, nJ, dJv, flg_porJ(flags[(c - sj) >> 3]), f);
#pragma unroll
            for (int l = 0; l < 4; ++l) acc[l] = fk[l] + f[l];
        }
        // ---- i face (i | i+1); the face (i-1 | i) comes from lane-1
        {
            adf_real8 nI[3], dIv[3]; double f[4];
            vm_ld3(sI, c, nb8, nI); vm_ld3(dI, c, nb8, dIv);
            const VmCell qR = vm_dn1(q0);
            vm_face<false>(K, gs, q0, qR, nI, dIv, flg_porI((uint8_t)flag0), f);
#pragma unroll
            for (int l = 0; l < 4; ++l) acc[l] += lane_up1(f[l]) - f[l];
        }
        // ---- j face (j | j+1)
        {
            adf_real8 nJ[3], dJv[3]; double f[4];
            const VmCell qjp = (row < VM_BY - 1) ? row_state(row + 1) : vm_ld(m, c + sj, gam, K.eddy);
            vm_ld3(sJ, c, nb8, nJ); vm_ld3(dJ, c, nb8, dJv);
            vm_face<false>(K, gs, q0, qjp, nJ, dJv, flg_porJ((uint8_t)flag0), f);
#pragma unroll
            for (int l = 0; l < 4; ++l) acc[l] -= f[l];
        }
        // ---- k face above the cell
        {
            adf_real8 nK[3], dKv[3]; double f[4];
            vm_ld3(sK, c, nb8, nK); vm_ld3(dK, c, nb8, dKv);
            vm_face<false>(K, gs, q0, qp1, nK, dKv, flg_porK((uint8_t)flag0), f);
#pragma unroll
            for (int l = 0; l < 4; ++l) { acc[l] -= f[l]; fk[l] = f[l]; }
        }
        if (out) {
            const double blank = flg_blank((uint8_t)flag0);
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const unsigned o = c + (l + 1) * nb8;
                double fwn = acc[l];
                if (FIRST) { stg(dw, o, fwn); continue; }
                if (kp.fwMode) {
                    fwn += ldg(fw, o);
                    stg(fw, o, fwn);
                }
                stg(dw, o, (ldg(dw, o) + fwn) * blank);
            }
            if (!FIRST && kp.fwMode) stg(dw, c, (ldg(dw, c) + ldg(fw, c)) * blank);    // the density residual has no viscous part
        }
        q0 = qp1;
        c += sk;
        __syncthreads();        // every wave has read the row states of this plane before they are replaced
    }
}

// ---------------------------------------------------------------------------
// FUSED nodal gradients + viscous fluxes, two workgroups per CU (tuning "viscous_tiled" = 2, the default).  The pair of round 2,
// k_node_grad_march and k_visc_march, exchanged the 12 nodal gradients through HBM (103 B per cell written, 125 B read back) and
// both read the state and the face normals: 194 + 427 B per cell by the counters, 0.45 + 0.84 ms.  Here ONE workgroup of four waves marches in k and keeps the
// gradients in LDS:
//   * wave r owns node row jn = j0-1+r of the tile: per plane it loads the raw values of the cell rows jn and jn+1 (no record
//     exchange between waves), forms the gradient of its node (i, jn, m-1) from the cell planes m-1 (carried) and m with the
//     factorised surface integral of k_node_grad_march, and writes it to a THREE-slot LDS ring: one barrier per plane, no
//     second one, because the slot written in step m+1 was last read in step m-1;
//   * with exactly the two cell rows it holds a wave can evaluate the i face and the k face of cell (i, jn, m-1) and the j face
//     ABOVE it, (jn | jn+1): three face evaluations per cell (3.25 with the tile-edge row of wave 0, which evaluates its upper
//     j face only), no state of a third row.  The flux through the j face BELOW the cell comes from wave r-1 through a
//     double-buffered LDS slot one step later (the barrier of the next plane orders it): a cell's sum is completed and stored
//     one plane behind its own faces;
//   * ring entries are [slot][node row][component pair][lane 1..61][2] (16-byte LDS accesses, consecutive lanes 16 bytes apart: no bank
//     conflict; with the twelve components of a lane side by side -- 96 bytes from lane to lane -- lanes n and n + 8 met in the same
//     banks); ring 70 272 B + hand-over 11 520 B =
//     81 792 B: two workgroups per CU, which run out of phase and hide each other's load latency (the single-workgroup
//     fusions of round 2 ran eight waves in barrier lockstep).  3 of 4 rows and 60 of 64 columns produce output.
// Reference: flowUtils.F90:1676-2026 (allNodalGradients), fluxes.F90:2534-3485 (viscousFlux), residuals.F90:334-344.
// STG: the gradients are also stored to b.grad (updateIntermed copy-out, wall stress).  FIRST: as k_visc_march.
// ---------------------------------------------------------------------------
#define GF_OUT 60
#define GF_ROWS 3
#define GF_NL 61                  // node columns kept per row: lanes 1..61
#define GF_G (12 * GF_NL)         // doubles of one node row of one plane
#define GF_RING (3 * 4 * GF_G)
#define GF_FJ (3 * GF_OUT * 4)    // doubles of one parity of the j-flux hand-over: rows 0..2, lanes 2..61, 4 components
#define GF_NW 4                   // waves (node rows) of a workgroup
#ifdef ADF_AD_BUILD
#define GF_MINWG 1                // dual numbers: the ring alone is 140 KB, the whole register file
#else
#define GF_MINWG 2
#endif

struct GfPtrs {
    GPTR(const double) w0; GPTR(const double) w1; GPTR(const double) w2; GPTR(const double) w3; GPTR(const double) p;
    GPTR(const double) rlv; GPTR(const double) rev; GPTR(const adf_real8) vol;
    GPTR(const adf_real8) sI; GPTR(const adf_real8) sJ; GPTR(const adf_real8) sK;
    unsigned nb8;
};

// state + volume of one cell
struct GfRaw { VmCell q; adf_real8 vol; };

__device__ __forceinline__ GfRaw gf_ld(const GfPtrs& m, unsigned o, double gam, bool eddy)
{
    GfRaw r;
    r.q.u = ldg(m.w1, o); r.q.v = ldg(m.w2, o); r.q.w = ldg(m.w3, o);
    r.q.na = -(gam * ldg(m.p, o)) * rcp_nr(ldg(m.w0, o));
    r.q.rlv = ldg(m.rlv, o);
    r.q.rev = eddy ? ldg(m.rev, o) : 0.0;
    r.vol = ldg(m.vol, o);
    return r;
}

// metric sums of one cell plane around the node column of the thread (the t-part of NgPlane)
struct GfMet { adf_real8 Pt[3], Q0t[3], Q1t[3], RIt[3], V; };

// two doubles moved as one 16-byte LDS access
struct __attribute__((aligned(16))) Dbl2 { double x, y; };
__device__ __forceinline__ Dbl2 mk2(double x, double y) { Dbl2 v; v.x = x; v.y = y; return v; }

// one cell plane of the rows jn, jn+1 as loaded
struct GfReq { double au, av, aw, ap, ar, alv, aev, bu, bv, bw, bp, br, blv, bev; adf_real8 avol, bvol, aI[3], aJm[3], aJ[3], aK[3], bI[3], bJ[3], bK[3]; };

template <bool QCR, bool FIRST, bool STG>
__global__ __launch_bounds__(64 * GF_NW, GF_MINWG) void k_visc_gf(const BlkView* __restrict__ tab, const int4* __restrict__ tiles, KParams kp)
{
    constexpr int NW = GF_NW, NSLOT = 3;
    __shared__ __attribute__((aligned(16))) double ring[NSLOT * NW * GF_G];   // [slot][node row 0..NW-1 = rows j0-1 .. j0+NW-2][component pair][lane-1][2]
    __shared__ __attribute__((aligned(16))) double fjx[2 * (NW - 1) * GF_OUT * 4];  // [parity][row][lane-2][component]
    constexpr int FJ = (NW - 1) * GF_OUT * 4;
    const int4 tl = tiles[blockIdx.x];
    if (tl.x < 0) return;
    const BlkView& b = tab[tl.x];
    const int lane = threadIdx.x, r = threadIdx.y;
    const int bx = tl.y & 0xffff, by = tl.y >> 16;
    const int i = bx * GF_OUT + lane;             // cells i0-2 .. i0+61, i0 = 2 + 60 bx
    const int j0 = 2 + by * (NW - 1);             // first produced cell row
    const int k0 = tl.z, k1 = tl.w;               // planes of the chunk
    const int jn = j0 - 1 + r;                    // node row of the wave; waves 1..3: also its cell row
    const int ic = (i < b.ib) ? i : b.ib;
    const int jA = (jn < b.jb) ? jn : b.jb, jB = (jn + 1 < b.jb) ? jn + 1 : b.jb;
    const bool outC = (r >= 1 && lane >= 2 && lane <= 61 && i <= b.il && jn <= b.jl);
    const bool outN = (lane >= 1 && lane <= 61 && i <= b.il && jn <= b.jl);
    const bool ringLane = (lane >= 1 && lane <= GF_NL);
    const int nl = ringLane ? lane - 1 : 0;       // ring column of the thread (clamped: the lanes outside read entry 0 and drop it)
    const int fl = (lane >= 2 && lane <= 61) ? lane - 2 : 0;
    const long nb = b.nbox;
    const unsigned nb8 = 8u * (unsigned)nb, sj = 8u * (unsigned)b.ldi, sk = 8u * (unsigned)b.ldk;
    GfPtrs m;
    m.w0 = (GPTR(const double))b.w; m.w1 = m.w0 + nb; m.w2 = m.w1 + nb; m.w3 = m.w2 + nb;
    m.p = (GPTR(const double))b.p; m.rlv = (GPTR(const double))b.rlv; m.rev = (GPTR(const double))b.rev;
    m.vol = (GPTR(const adf_real8))b.vol;
    m.sI = (GPTR(const adf_real8))b.sI; m.sJ = (GPTR(const adf_real8))b.sJ; m.sK = (GPTR(const adf_real8))b.sK;
    m.nb8 = nb8;
    GPTR(const adf_real8) xcen = (GPTR(const adf_real8))b.xc;
    GPTR(const uint8_t) flags = (GPTR(const uint8_t))b.flags;
    GPTR(double) dw = (GPTR(double))b.dw;
    GPTR(double) fw = (GPTR(double))b.fw;
    GPTR(double) grad = (GPTR(double))b.grad;
    VmK K;
    K.porV = 0.5 * kp.rFil; K.eddy = kp.eddyModel != 0;
    K.hl = 1.0 / (kp.prandtl * (kp.gammaConstant - 1.0)); K.ht = 1.0 / (kp.prandtlTurb * (kp.gammaConstant - 1.0));
    const adf_real8 gam = kp.gammaConstant;
    // byte offsets of the cells (ic, jA, m) and (ic, jB, m); the row below the own one (jn >= 1)
    unsigned cA = 8u * (unsigned)(ic + jA * b.ldi + (k0 - 1) * b.ldk);
    const unsigned dB = 8u * (unsigned)((jB - jA) * b.ldi);          // row jn+1 relative to row jn (0 at the upper end of the box)
    const unsigned dM = (jA >= 1) ? sj : 0u;                          // row jn-1
    // carried: state of the rows jn, jn+1 at the previous plane, metric sums and normals of that plane, the k-face flux, the own
    // part of the flux sum of the plane before
    VmCell qA, qB;
    GfMet S;
    adf_real8 sKA[3], sKB[3];
    double fk[4], pend[4];
    adf_real8 xcP[3] = {0.0, 0.0, 0.0};        // centre of cell (i, jn, mm-1)
    int flagP = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) { sKA[d] = ldg(m.sK, cA - sk + d * nb8); sKB[d] = ldg(m.sK, cA + dB - sk + d * nb8); }
#pragma unroll
    for (int l = 0; l < 4; ++l) { fk[l] = 0.0; pend[l] = 0.0; }
    qA.u = qA.v = qA.w = qA.na = qA.rlv = qA.rev = 0.0;
    qB = qA;
    S.V = 0.0;
#pragma unroll
    for (int d = 0; d < 3; ++d) S.Pt[d] = S.Q0t[d] = S.Q1t[d] = S.RIt[d] = 0.0;
    // completes the flux sum of cell plane kc (byte offset c) with the j flux handed over by the wave below and stores it
    auto finish = [&](unsigned c, const double* __restrict__ fjr, int flg) {
        const Dbl2 f01 = *reinterpret_cast<const Dbl2*>(fjr), f23 = *reinterpret_cast<const Dbl2*>(fjr + 2);
        const double fl4[4] = {f01.x, f01.y, f23.x, f23.y};
        if (!outC) return;
        const adf_real8 blank = flg_blank((uint8_t)flg);
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const unsigned o = c + (l + 1) * nb8;
            double fwn = pend[l] + fl4[l];
            if (FIRST) { stg(dw, o, fwn); continue; }
            if (kp.fwMode) {
                fwn += ldg(fw, o);
                stg(fw, o, fwn);
            }
            stg(dw, o, (ldg(dw, o) + fwn) * blank);
        }
        if (!FIRST && kp.fwMode) stg(dw, c, (ldg(dw, c) + ldg(fw, c)) * blank);    // the density residual has no viscous part
    };
    auto request = [&](unsigned c) {
        GfReq q;
        q.au = ldg(m.w1, c); q.av = ldg(m.w2, c); q.aw = ldg(m.w3, c); q.ap = ldg(m.p, c); q.ar = ldg(m.w0, c);
        q.alv = ldg(m.rlv, c); q.aev = K.eddy ? ldg(m.rev, c) : 0.0; q.avol = ldg(m.vol, c);
        const unsigned cb = c + dB;
        q.bu = ldg(m.w1, cb); q.bv = ldg(m.w2, cb); q.bw = ldg(m.w3, cb); q.bp = ldg(m.p, cb); q.br = ldg(m.w0, cb);
        q.blv = ldg(m.rlv, cb); q.bev = K.eddy ? ldg(m.rev, cb) : 0.0; q.bvol = ldg(m.vol, cb);
        vm_ld3(m.sI, c, nb8, q.aI); vm_ld3(m.sJ, c - dM, nb8, q.aJm); vm_ld3(m.sJ, c, nb8, q.aJ); vm_ld3(m.sK, c, nb8, q.aK);
        vm_ld3(m.sI, cb, nb8, q.bI); vm_ld3(m.sJ, cb, nb8, q.bJ); vm_ld3(m.sK, cb, nb8, q.bK);
        return q;
    };
    for (int mm = k0 - 1; mm <= k1 + 1; ++mm) {
        const unsigned cF = cA - sk;
        const bool facePlane = (mm >= k0);
        const bool full = (mm > k0);                           // all faces (first step of the march: the k face below plane k0 only)
        // the vectors between cell centres (fluxes.F90:2673-2690, 2966-2983, 3260-3277) are differences of the stored centres: of the
        // own cell at this plane (kept for the next step) and of the row above at the plane of the faces -- 24 unique bytes per cell
        // where dI / dJ / dK were 72
        adf_real8 xcN[3], xcB[3], sIA[3], sJA[3];
        int flag0 = 0;
        auto face_loads = [&]() {
            vm_ld3(xcen, cA, nb8, xcN);
            if (!facePlane) return;
            flag0 = flags[cF >> 3];
            if (full) {
                vm_ld3(xcen, cF + dB, nb8, xcB);
                if (r >= 1) vm_ld3(m.sI, cF, nb8, sIA);       // (sI / sJ of that plane again: carried they spill -- 16 B of scratch cost 0.02 ms, profiles/r06_g_ab.txt)
                vm_ld3(m.sJ, cF, nb8, sJA);
            }
        };
        // ---- cell plane mm of the rows jn and jn+1: state, normals, volume; centre-to-centre vectors and flags of plane mm-1
        const GfReq cur = request(cA);
        GfRaw a, bq;
        a.q.u = cur.au; a.q.v = cur.av; a.q.w = cur.aw; a.q.na = -(gam * cur.ap) * rcp_nr(cur.ar); a.q.rlv = cur.alv; a.q.rev = cur.aev; a.vol = cur.avol;
        bq.q.u = cur.bu; bq.q.v = cur.bv; bq.q.w = cur.bw; bq.q.na = -(gam * cur.bp) * rcp_nr(cur.br); bq.q.rlv = cur.blv; bq.q.rev = cur.bev; bq.vol = cur.bvol;
        adf_real8 aI[3], aJm[3], aJ[3], aK[3], bI[3], bJ[3], bK[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            aI[d] = cur.aI[d]; aJm[d] = cur.aJm[d]; aJ[d] = cur.aJ[d]; aK[d] = cur.aK[d]; bI[d] = cur.bI[d]; bJ[d] = cur.bJ[d]; bK[d] = cur.bK[d];
        }
        // ---- metric sums of this plane (ng_finish for both rows; the row above takes sJ(j-1) from the own row)
        GfMet N;
        {
            adf_real8 tJa[3], tKa[3], tJb[3], tKb[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const adf_real8 tIa = lane_up1(aI[d]) + aI[d], tIb = lane_up1(bI[d]) + bI[d];
                N.RIt[d] = tIa + tIb;
                tJa[d] = aJm[d] + aJ[d]; tJb[d] = aJ[d] + bJ[d];
                tKa[d] = sKA[d] + aK[d]; tKb[d] = sKB[d] + bK[d];
            }
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                N.Q0t[d] = tJa[d] + lane_dn1(tJa[d]);
                N.Q1t[d] = tJb[d] + lane_dn1(tJb[d]);
                N.Pt[d] = (tKa[d] + lane_dn1(tKa[d])) + (tKb[d] + lane_dn1(tKb[d]));
            }
            N.V = (a.vol + lane_dn1(a.vol)) + (bq.vol + lane_dn1(bq.vol));
        }
        // ---- gradient of node (i, jn, mm-1) from the cell planes mm-1 (qA, qB, S) and mm -> ring
        if (mm >= k0) {
            const double sA[4] = {qA.u, qA.v, qA.w, qA.na}, sB[4] = {qB.u, qB.v, qB.w, qB.na};
            const double nA[4] = {a.q.u, a.q.v, a.q.w, a.q.na}, nB[4] = {bq.q.u, bq.q.v, bq.q.w, bq.q.na};
            double g[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) g[q] = 0.0;
            double SQ0[4], SQ1[4], NQ0[4], NQ1[4], ph[4];
            adf_real8 t3[3];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                SQ0[v] = sA[v] + lane_dn1(sA[v]); SQ1[v] = sB[v] + lane_dn1(sB[v]);
                NQ0[v] = nA[v] + lane_dn1(nA[v]); NQ1[v] = nB[v] + lane_dn1(nB[v]);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) ph[v] = SQ0[v] + SQ1[v];
            ng_outer<true>(g, ph, S.Pt);                          // k direction: below the node -, above +
#pragma unroll
            for (int v = 0; v < 4; ++v) ph[v] = NQ0[v] + NQ1[v];
            ng_outer<false>(g, ph, N.Pt);
#pragma unroll
            for (int d = 0; d < 3; ++d) t3[d] = S.Q0t[d] + N.Q0t[d];
#pragma unroll
            for (int v = 0; v < 4; ++v) ph[v] = SQ0[v] + NQ0[v];
            ng_outer<true>(g, ph, t3);                            // j direction: own row -, row above +
#pragma unroll
            for (int d = 0; d < 3; ++d) t3[d] = S.Q1t[d] + N.Q1t[d];
#pragma unroll
            for (int v = 0; v < 4; ++v) ph[v] = SQ1[v] + NQ1[v];
            ng_outer<false>(g, ph, t3);
#pragma unroll
            for (int d = 0; d < 3; ++d) t3[d] = S.RIt[d] + N.RIt[d];
#pragma unroll
            for (int v = 0; v < 4; ++v) ph[v] = (sA[v] + sB[v]) + (nA[v] + nB[v]);
            ng_outer<true>(g, ph, t3);                            // i direction: own column -, column i+1 +
            adf_real8 t1[3];
            double ph1[4];
#pragma unroll
            for (int d = 0; d < 3; ++d) t1[d] = lane_dn1(t3[d]);
#pragma unroll
            for (int v = 0; v < 4; ++v) ph1[v] = lane_dn1(ph[v]);
            ng_outer<false>(g, ph1, t1);
            // a QUARTER of the gradient goes to the ring (the faces then average four nodes by adding); with the 0.25 of the surface
            // integral: 1 / (16 V).  Powers of two: the face gradients are bitwise what 0.25 (g0 + g1 + g2 + g3) gives
            const adf_real8 oneOverV = 0.0625 * rcp_nr(S.V + N.V);
#pragma unroll
            for (int q = 0; q < 12; ++q) g[q] *= oneOverV;
            if (ringLane) {
                double* __restrict__ xo = ring + (((mm - k0) % NSLOT) * NW + r) * GF_G + nl * 2;     // node plane mm-1 -> slot (mm-k0) % NSLOT
#pragma unroll
                for (int q = 0; q < 12; q += 2) *reinterpret_cast<Dbl2*>(xo + q * GF_NL) = mk2(g[q], g[q + 1]);
            }
            if (STG && outN) {
#pragma unroll
                for (int q = 0; q < 12; ++q) stg(grad + q * nb, cF, 4.0 * g[q]);
            }
        }
        // ---- loads of the face part (cell plane mm-1), requested above the barrier; sI / sJ of that plane again (carried they spill)
        face_loads();          // requested above the barrier (at the top of the step: no faster, profiles/r03_f)
        __syncthreads();
        if (facePlane) {
            const double* __restrict__ xb = ring + (((mm - k0) % NSLOT) * NW) * GF_G + nl * 2;                    // node plane mm-1
            const double* __restrict__ xp = ring + (((mm - k0 + NSLOT - 1) % NSLOT) * NW) * GF_G + nl * 2;        // node plane mm-2
            const int oM = (r >= 1 ? r - 1 : 0) * GF_G, o0 = r * GF_G;                                // node rows jn-1 and jn
            if (full && r >= 1 && mm - 2 >= k0) {
                // cell plane mm-2: the j flux from the wave below has arrived (written in step mm-1)
                finish(cF - sk, fjx + ((mm - 1) & 1) * FJ + ((r - 1) * GF_OUT + fl) * 4, flagP);
            }
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            if (full) {
                // ---- j face (jn | jn+1): nodes (i-1..i, jn, mm-2..mm-1); handed to the wave above
                double gs[12], f[4];
#pragma unroll
                for (int q = 0; q < 12; q += 2) {
                    const Dbl2 u = *reinterpret_cast<const Dbl2*>(xp + o0 + q * GF_NL), v = *reinterpret_cast<const Dbl2*>(xb + o0 + q * GF_NL);
                    const double s0 = u.x + v.x, s1 = u.y + v.y;
                    gs[q] = s0 + lane_up1(s0); gs[q + 1] = s1 + lane_up1(s1);
                }
                const adf_real8 dJv[3] = {xcB[0] - xcP[0], xcB[1] - xcP[1], xcB[2] - xcP[2]};
                vm_face<QCR>(K, gs, qA, qB, sJA, dJv, flg_porJ((uint8_t)flag0), f);
#pragma unroll
                for (int l = 0; l < 4; ++l) acc[l] = -f[l];
                if (r < NW - 1 && lane >= 2 && lane <= 61) {
                    double* __restrict__ fo = fjx + (mm & 1) * FJ + (r * GF_OUT + fl) * 4;
                    *reinterpret_cast<Dbl2*>(fo) = mk2(f[0], f[1]);
                    *reinterpret_cast<Dbl2*>(fo + 2) = mk2(f[2], f[3]);
                }
            }
            if (r >= 1) {
                if (full) {
                    // ---- i face (i | i+1): nodes (i, jn-1..jn, mm-2..mm-1); the face (i-1 | i) comes from lane-1
                    double gs[12], f[4];
#pragma unroll
                    for (int q = 0; q < 12; q += 2) {
                        const Dbl2 u = *reinterpret_cast<const Dbl2*>(xp + oM + q * GF_NL), v = *reinterpret_cast<const Dbl2*>(xp + o0 + q * GF_NL);
                        const Dbl2 w = *reinterpret_cast<const Dbl2*>(xb + oM + q * GF_NL), z = *reinterpret_cast<const Dbl2*>(xb + o0 + q * GF_NL);
                        gs[q] = (u.x + v.x) + (w.x + z.x); gs[q + 1] = (u.y + v.y) + (w.y + z.y);
                    }
                    const VmCell qR = vm_dn1(qA);
                    const adf_real8 dIv[3] = {lane_dn1(xcP[0]) - xcP[0], lane_dn1(xcP[1]) - xcP[1], lane_dn1(xcP[2]) - xcP[2]};
                    vm_face<QCR>(K, gs, qA, qR, sIA, dIv, flg_porI((uint8_t)flag0), f);
#pragma unroll
                    for (int l = 0; l < 4; ++l) acc[l] += lane_up1(f[l]) - f[l];
                }
                // ---- k face above cell plane mm-1: nodes (i-1..i, jn-1..jn, mm-1); sKA still holds sK of plane mm-1
                {
                    double gs[12], f[4];
#pragma unroll
                    for (int q = 0; q < 12; q += 2) {
                        const Dbl2 w = *reinterpret_cast<const Dbl2*>(xb + oM + q * GF_NL), z = *reinterpret_cast<const Dbl2*>(xb + o0 + q * GF_NL);
                        const double s0 = w.x + z.x, s1 = w.y + z.y;
                        gs[q] = s0 + lane_up1(s0); gs[q + 1] = s1 + lane_up1(s1);
                    }
                    const adf_real8 dKv[3] = {xcN[0] - xcP[0], xcN[1] - xcP[1], xcN[2] - xcP[2]};
                    vm_face<QCR>(K, gs, qA, a.q, sKA, dKv, flg_porK((uint8_t)flag0), f);
#pragma unroll
                    for (int l = 0; l < 4; ++l) { pend[l] = (acc[l] + fk[l]) - f[l]; fk[l] = f[l]; }
                }
                flagP = flag0;
            }
        }
        // ---- advance
        qA = a.q; qB = bq.q;
        S = N;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            sKA[d] = aK[d]; sKB[d] = bK[d];
            xcP[d] = xcN[d];
        }
        cA += sk;
    }
    // ---- the last plane of the chunk: its j flux was handed over in the last step
    __syncthreads();
    if (r >= 1 && k1 >= k0) finish(cA - 2 * sk, fjx + ((k1 + 1) & 1) * FJ + ((r - 1) * GF_OUT + fl) * 4, flagP);
}

// ---------------------------------------------------------------------------
// viscousFluxApprox (fluxes.F90:3487-3859): thin-layer form for the preconditioner assembly.  The gradient on a
// face is the difference of the two cell values along the centre-to-centre vector d (no nodal gradients):
// grad(q) = (q_R - q_L) d / |d|^2.  d is the static face vector of k_face_vectors (same node order).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void visc_face_approx(const KParams& kp, const VCell& L, const VCell& R, const double fN[3],
                                                 const double dN[3], int por_code, double sign, double acc[5])
{
    double ss = 1.0 / (dN[0] * dN[0] + dN[1] * dN[1] + dN[2] * dN[2]);
    const double ssx = ss * dN[0], ssy = ss * dN[1], ssz = ss * dN[2];
    double dd;
    dd = R.u - L.u;
    const double u_x = dd * ssx, u_y = dd * ssy, u_z = dd * ssz;
    dd = R.v - L.v;
    const double v_x = dd * ssx, v_y = dd * ssy, v_z = dd * ssz;
    dd = R.w - L.w;
    const double w_x = dd * ssx, w_y = dd * ssy, w_z = dd * ssz;
    dd = R.aa - L.aa;
    double q_x = -dd * ssx, q_y = -dd * ssy, q_z = -dd * ssz;
    double por = 0.5 * kp.rFil;
    if (por_code == ADF_POR_NOFLUX) por = 0.0;
    const double mul = por * (L.rlv + R.rlv);
    double mue = 0.0;
    if (kp.eddyModel) mue = por * (L.rev + R.rev);
    const double mut = mul + mue;
    const double gm1 = 0.5 * (L.gam + R.gam) - 1.0;
    const double factLamHeat = 1.0 / (kp.prandtl * gm1), factTurbHeat = 1.0 / (kp.prandtlTurb * gm1);
    const double heatCoef = mul * factLamHeat + mue * factTurbHeat;
    const double fracDiv = (2.0 * (1.0 / 3.0)) * (u_x + v_y + w_z);
    const double tauxx = mut * (2.0 * u_x - fracDiv), tauyy = mut * (2.0 * v_y - fracDiv), tauzz = mut * (2.0 * w_z - fracDiv);
    const double tauxy = mut * (u_y + v_x), tauxz = mut * (u_z + w_x), tauyz = mut * (v_z + w_y);
    q_x = heatCoef * q_x; q_y = heatCoef * q_y; q_z = heatCoef * q_z;
    const double ubar = 0.5 * (L.u + R.u), vbar = 0.5 * (L.v + R.v), wbar = 0.5 * (L.w + R.w);
    const double nx = fN[0], ny = fN[1], nz = fN[2];
    const double fmx = tauxx * nx + tauxy * ny + tauxz * nz;
    const double fmy = tauxy * nx + tauyy * ny + tauyz * nz;
    const double fmz = tauxz * nx + tauyz * ny + tauzz * nz;
    const double frhoE = (ubar * tauxx + vbar * tauxy + wbar * tauxz) * nx + (ubar * tauxy + vbar * tauyy + wbar * tauyz) * ny +
                         (ubar * tauxz + vbar * tauyz + wbar * tauzz) * nz - q_x * nx - q_y * ny - q_z * nz;
    acc[1] += sign * fmx;
    acc[2] += sign * fmy;
    acc[3] += sign * fmz;
    acc[4] += sign * frhoE;
}

__global__ __launch_bounds__(VS_BX* VS_BY) void k_viscous_approx(BlkView b, KParams kp)
{
    const int i = blockIdx.x * VS_BX + threadIdx.x + 2;
    const int j = blockIdx.y * VS_BY + threadIdx.y + 2;
    const int k = blockIdx.z + 2;
    if (i > b.il || j > b.jl) return;
    const long c = b.idx(i, j, k), nb = b.nbox;
    const long si = 1, sj = b.ldi, sk = b.ldk;
    const uint8_t f0 = b.flags[c];
    double acc[5] = {0, 0, 0, 0, 0};
    const VCell C = vcell_at(b, kp, c);
    // reference sweep order i, j, k (fluxes.F90:3520, 3666, 3762)
    const long sd3[3] = {si, sj, sk};
    const adf_real8* sN3[3] = {b.sI, b.sJ, b.sK};
    const adf_real8* dN3[3] = {b.dI, b.dJ, b.dK};
    const int shift3[3] = {0, 2, 4};
#pragma unroll 1
    for (int d = 0; d < 3; ++d) {
        const long sd = sd3[d], cm = c - sd;
        const adf_real8* __restrict__ sN = sN3[d];
        const adf_real8* __restrict__ dN = dN3[d];
        const VCell M = vcell_at(b, kp, cm), P = vcell_at(b, kp, c + sd);
        const double nM[3] = {sN[cm], sN[cm + nb], sN[cm + 2 * nb]}, nP[3] = {sN[c], sN[c + nb], sN[c + 2 * nb]};
        const double dM[3] = {dN[cm], dN[cm + nb], dN[cm + 2 * nb]}, dP[3] = {dN[c], dN[c + nb], dN[c + 2 * nb]};
        visc_face_approx(kp, M, C, nM, dM, (b.flags[cm] >> shift3[d]) & 3, +1.0, acc);
        visc_face_approx(kp, C, P, nP, dP, (f0 >> shift3[d]) & 3, -1.0, acc);
    }
    const double blank = flg_blank(f0);
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        const double fwn = (kp.fwMode ? b.fw[c + l * nb] : 0.0) + acc[l];   // without fwMode dw already holds dw + fw
        if (kp.fwMode) b.fw[c + l * nb] = fwn;
        b.dw[c + l * nb] = (b.dw[c + l * nb] + fwn) * blank;
    }
}

void launch_viscous_approx(const BlkView& b, const KParams& kp, hipStream_t s)
{
    dim3 blk(VS_BX, VS_BY, 1);
    dim3 gc((b.nx + VS_BX - 1) / VS_BX, (b.ny + VS_BY - 1) / VS_BY, b.nz);
    hipLaunchKernelGGL(k_viscous_approx, gc, blk, 0, s, b, kp);
}

// gather forms, one block per launch (tuning "viscous_tiled" = 0, kept for A/B measurements)
void launch_viscous(const BlkView& b, const KParams& kp, hipStream_t s)
{
    dim3 blk(VS_BX, VS_BY, 1);
    dim3 gn((b.il + 15 + VS_BX - 1) / VS_BX, (b.jl + VS_BY - 1) / VS_BY, b.kl);
    hipLaunchKernelGGL(k_nodal_gradients, gn, blk, 0, s, b);
    dim3 gc((b.nx + VS_BX - 1) / VS_BX, (b.ny + VS_BY - 1) / VS_BY, b.nz);
    hipLaunchKernelGGL(k_viscous, gc, blk, 0, s, b, kp);
}

// marching face-flux kernel over the tile table of the level (tuning viscous_tiled >= 2)
// viscousFluxApprox of every block of the level (thin-layer form, no nodal gradients)
void launch_visc_march_approx(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s)
{
    if (ntiles <= 0) return;
    if (kp.viscFirst) hipLaunchKernelGGL((k_visc_approx_march<true>), dim3(ntiles), dim3(64, VM_BY, 1), 0, s, tab, tiles, kp, ::g_march_kch);
    else hipLaunchKernelGGL((k_visc_approx_march<false>), dim3(ntiles), dim3(64, VM_BY, 1), 0, s, tab, tiles, kp, ::g_march_kch);
}

// fused nodal gradients + viscous fluxes over the level's round-fitted chunk table (api.hip ensure_gf_tiles)
void launch_visc_gf(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, bool storeGrad, hipStream_t s)
{
    if (ntiles <= 0) return;
    const dim3 grd(ntiles), blk(64, GF_NW, 1);
#define GF_LAUNCH(Q, F, G) hipLaunchKernelGGL((k_visc_gf<Q, F, G>), grd, blk, 0, s, tab, tiles, kp)
    if (kp.useQCR) {
        if (kp.viscFirst) { if (storeGrad) GF_LAUNCH(true, true, true); else GF_LAUNCH(true, true, false); }
        else { if (storeGrad) GF_LAUNCH(true, false, true); else GF_LAUNCH(true, false, false); }
    } else {
        if (kp.viscFirst) { if (storeGrad) GF_LAUNCH(false, true, true); else GF_LAUNCH(false, true, false); }
        else { if (storeGrad) GF_LAUNCH(false, false, true); else GF_LAUNCH(false, false, false); }
    }
#undef GF_LAUNCH
}

#ifndef ADF_AD_BUILD
int g_sa_march = 1;         // tuning "sa_march": 0 = gather kernel (k_sa_residual), 1 = k-march (kernels_sa_march.hip)

int viscous_is_tiled() { return g_viscous_tiled; }
#endif
