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
#include "internal.h"

#define VS_BX 64
#define VS_BY 4

__device__ __forceinline__ double aa_at(const BlkView& b, long q) { return b.gamma[q] * b.p[q] / b.w[q]; }

// one integration point of the dual-cell surface integral (flowUtils.F90:1712-1791):
// cells c0, c0+s1, c0+s2, c0+s1+s2 (a 2x2 patch normal to direction d) and the
// normals of the faces below (c-sd) and above (c) each of those four cells.
__device__ __forceinline__ void grad_point(const BlkView& b, long c0, long sd, long s1, long s2,
                                           const double* __restrict__ sN, double sign, double g[12])
{
    const long nb = b.nbox;
    const long cc[4] = {c0, c0 + s1, c0 + s2, c0 + s1 + s2};
    double sx = 0.0, sy = 0.0, sz = 0.0;
    // reference summation order: the four faces at index-1 first, then the four at index
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        sx += sN[cc[m] - sd];
        sy += sN[cc[m] - sd + nb];
        sz += sN[cc[m] - sd + 2 * nb];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        sx += sN[cc[m]];
        sy += sN[cc[m] + nb];
        sz += sN[cc[m] + 2 * nb];
    }
    const double ubar = 0.25 * (b.w[cc[0] + nb] + b.w[cc[1] + nb] + b.w[cc[2] + nb] + b.w[cc[3] + nb]);
    const double vbar = 0.25 * (b.w[cc[0] + 2 * nb] + b.w[cc[1] + 2 * nb] + b.w[cc[2] + 2 * nb] + b.w[cc[3] + 2 * nb]);
    const double wbar = 0.25 * (b.w[cc[0] + 3 * nb] + b.w[cc[1] + 3 * nb] + b.w[cc[2] + 3 * nb] + b.w[cc[3] + 3 * nb]);
    const double a2 = 0.25 * (aa_at(b, cc[0]) + aa_at(b, cc[1]) + aa_at(b, cc[2]) + aa_at(b, cc[3]));
    g[0] += sign * ubar * sx; g[1] += sign * ubar * sy; g[2] += sign * ubar * sz;
    g[3] += sign * vbar * sx; g[4] += sign * vbar * sy; g[5] += sign * vbar * sz;
    g[6] += sign * wbar * sx; g[7] += sign * wbar * sy; g[8] += sign * wbar * sz;
    g[9] -= sign * a2 * sx; g[10] -= sign * a2 * sy; g[11] -= sign * a2 * sz;
}

// nodes 1..il x 1..jl x 1..kl ; node (i,j,k) is stored at cell index (i,j,k)
__global__ __launch_bounds__(VS_BX* VS_BY) void k_nodal_gradients(BlkView b)
{
    const int i = blockIdx.x * VS_BX + threadIdx.x + (2 - 16);   // aligned rows, see internal.h
    const int j = blockIdx.y * VS_BY + threadIdx.y + 1;
    const int k = blockIdx.z + 1;
    if (i < 1 || i > b.il || j > b.jl) return;
    const long c = b.idx(i, j, k);
    const long si = 1, sj = b.ldi, sk = b.ldk;
    double g[12];
#pragma unroll
    for (int m = 0; m < 12; ++m) g[m] = 0.0;
    // k-direction: point k gives "-", point k+1 gives "+" (flowUtils.F90:1759-1791)
    grad_point(b, c, sk, si, sj, b.sK, -1.0, g);
    grad_point(b, c + sk, sk, si, sj, b.sK, +1.0, g);
    // j-direction (cells (i..i+1, j, k..k+1); flowUtils.F90:1805-1892)
    grad_point(b, c, sj, si, sk, b.sJ, -1.0, g);
    grad_point(b, c + sj, sj, si, sk, b.sJ, +1.0, g);
    // i-direction (cells (i, j..j+1, k..k+1); flowUtils.F90:1894-1979)
    grad_point(b, c, si, sj, sk, b.sI, -1.0, g);
    grad_point(b, c + si, si, sj, sk, b.sI, +1.0, g);
    const double oneOverV = 1.0 / (b.vol[c] + b.vol[c + sk] + b.vol[c + si] + b.vol[c + si + sk] + b.vol[c + sj] +
                                   b.vol[c + sj + sk] + b.vol[c + si + sj] + b.vol[c + si + sj + sk]);
#pragma unroll
    for (int m = 0; m < 12; ++m) b.grad[c + m * b.nbox] = g[m] * oneOverV;
}

// viscous flux through the face between cells cL and cL+sd (fluxes.F90:2610-2860);
// the four face nodes are cL, cL-s1, cL-s2, cL-s1-s2.
__device__ __forceinline__ void visc_face(const BlkView& b, const KParams& kp, long cL, long sd, long s1, long s2,
                                          const double* __restrict__ sN, int por_code, double sign, double acc[5])
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
        const double* xx = b.x + d * nb;
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
    const double* sN3[3] = {b.sK, b.sJ, b.sI};
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
        const double fwn = b.fw[c + l * nb] + acc[l];
        if (kp.fwMode) b.fw[c + l * nb] = fwn;
        b.dw[c + l * nb] = (b.dw[c + l * nb] + fwn) * blank;
    }
}

void launch_viscous(const BlkView& b, const KParams& kp, hipStream_t s)
{
    dim3 blk(VS_BX, VS_BY, 1);
    dim3 gn((b.il + 15 + VS_BX - 1) / VS_BX, (b.jl + VS_BY - 1) / VS_BY, b.kl);
    hipLaunchKernelGGL(k_nodal_gradients, gn, blk, 0, s, b);
    dim3 gc((b.nx + VS_BX - 1) / VS_BX, (b.ny + VS_BY - 1) / VS_BY, b.nz);
    hipLaunchKernelGGL(k_viscous, gc, blk, 0, s, b, kp);
}
