// Flow boundary conditions on the device (SURVEY.md §8f "next" row 1).
//
// Reference semantics (src/solver/BCRoutines.F90):
//   applyAllBC_block      :57-221   order of the kinds, subfaces in index order
//   bcSymm1stHalo/2ndHalo :223-330
//   bcSymmPolar1st/2ndHalo:332-487
//   bcSubsonicOutflow     :693-802  (also outflow mass bleeds)
//   bcSubsonicInflow      :804-1061 (total conditions / mass flow; cpConstant)
//   bcNSWallAdiabatic     :489-577
//   bcNSWallIsoThermal    :579-691
//   bcEulerWall           :1063-1280 (constant / linear pressure extrapolation)
//   bcFarfield            :1282-1409
//   bcSupersonicInflow    :1411-1477
//   bcExtrap              :1479-1570
//   computeEtot           :1816-1868 (cpConstant, no k correction: SA carries no k)
//   extrapolate2ndHalo    :1870-1918
//   setBCPointers         src/utils/utils.F90:881-1175 (ww0..ww3 = 2nd halo, 1st halo,
//                         1st and 2nd interior slab of the block face)
// One thread per cell (i,j) of the subface range icBeg..icEnd x jcBeg..jcEnd.  A subface
// is a 2-D slab: its cost is O(N^2) against the O(N^3) residual, so the kernels are
// written for clarity; lanes run along the first face index (i for j/k faces: coalesced).
// Subfaces that share halo cells along block edges read what earlier subfaces wrote, so
// they are separate launches in the reference's order.
#include <algorithm>
#include <vector>

#include "internal.h"

// Level-batched launches: the subfaces of every block of a level sit in one device array; a launch runs over a
// list of entries (blockIdx.y) that may execute concurrently - never two overlapping subfaces of one block.
#define BC_ARGS const BlkView* __restrict__ tab, const BcEntry* __restrict__ ent, const int* __restrict__ order
#define BC_PROLOGUE                                   \
    const BcEntry& e_ = ent[order[blockIdx.y]];       \
    const BlkView& b = tab[e_.slot];                  \
    const BcFaceDev& f = e_.f;

struct BcSlab {            // cell offsets of the four slabs and the BCData index of one (i,j)
    long c0, c1, c2, c3;   // 2nd halo, 1st halo, 1st interior, 2nd interior
    long f;                // (i - icBeg) + isize * (j - jcBeg)
    long fn;               // stride between components of norm / uSlip
};

__device__ __forceinline__ bool bc_slab_t(const BlkView& b, const BcFaceDev& f, long t, BcSlab& s)
{
    const int isize = f.icEnd - f.icBeg + 1, jsize = f.jcEnd - f.jcBeg + 1;
    if (t >= (long)isize * jsize) return false;
    const int i = f.icBeg + (int)(t % isize), j = f.jcBeg + (int)(t / isize);
    long base, step;
    switch (f.faceID) {
    case ADFLOW_IMIN: base = b.idx(0, i, j); step = 1; break;
    case ADFLOW_IMAX: base = b.idx(b.ib, i, j); step = -1; break;
    case ADFLOW_JMIN: base = b.idx(i, 0, j); step = b.ldi; break;
    case ADFLOW_JMAX: base = b.idx(i, b.jb, j); step = -(long)b.ldi; break;
    case ADFLOW_KMIN: base = b.idx(i, j, 0); step = b.ldk; break;
    default: base = b.idx(i, j, b.kb); step = -(long)b.ldk; break;
    }
    s.c0 = base; s.c1 = base + step; s.c2 = base + 2 * step; s.c3 = base + 3 * step;
    s.f = t;
    s.fn = (long)isize * jsize;
    return true;
}
__device__ __forceinline__ bool bc_slab(const BlkView& b, const BcFaceDev& f, BcSlab& s)
{
    return bc_slab_t(b, f, (long)blockIdx.x * blockDim.x + threadIdx.x, s);
}

// computeEtot (BCRoutines.F90:1816-1868)
__device__ __forceinline__ void bc_etot(const BlkView& b, const KParams& kp, long c)
{
    const long nb = b.nbox;
    const double ovgm1 = 1.0 / (kp.gammaConstant - 1.0);
    const double u = b.w[c + nb], v = b.w[c + 2 * nb], w = b.w[c + 3 * nb];
    b.w[c + 4 * nb] = ovgm1 * b.p[c] + 0.5 * b.w[c] * (u * u + v * v + w * w);
}

// extrapolate2ndHalo (BCRoutines.F90:1870-1918)
__device__ __forceinline__ void bc_second_halo(const BlkView& b, const KParams& kp, const BcSlab& s)
{
    const long nb = b.nbox;
    const double factor = 0.5;
    double r = 2.0 * b.w[s.c1] - b.w[s.c2];
    r = fmax(factor * b.w[s.c1], r);
    b.w[s.c0] = r;
#pragma unroll
    for (int l = 1; l <= 3; ++l) b.w[s.c0 + l * nb] = 2.0 * b.w[s.c1 + l * nb] - b.w[s.c2 + l * nb];
    b.p[s.c0] = fmax(factor * b.p[s.c1], 2.0 * b.p[s.c1] - b.p[s.c2]);
    if (kp.viscous) b.rlv[s.c0] = b.rlv[s.c1];
    if (kp.eddyModel) b.rev[s.c0] = b.rev[s.c1];
    bc_etot(b, kp, s.c0);
}

// First halo h (with its energy, computeEtot) and -- second -- the extrapolated second halo (extrapolate2ndHalo) written from
// REGISTERS: i2 = (rho, u, v, w, p) of the first interior slab.  The kernels above this line write the first halo and read it back for
// the energy and for the second halo (three dependent trips to memory per thread of a launch that fills an eighth of the chip); the
// common boundary kinds of a RANS mesh (farfield, walls, extrapolation) go through here.  Same expressions, same values.
struct BcState { double rho, u, v, w, p, rlv, rev; };
__device__ __forceinline__ void bc_store_halos(const BlkView& b, const KParams& kp, const BcSlab& s, const BcState& h, const BcState& i2,
                                               bool second)
{
    const long nb = b.nbox;
    const double ovgm1 = 1.0 / (kp.gammaConstant - 1.0);
    b.w[s.c1] = h.rho; b.w[s.c1 + nb] = h.u; b.w[s.c1 + 2 * nb] = h.v; b.w[s.c1 + 3 * nb] = h.w;
    b.p[s.c1] = h.p;
    if (kp.viscous) b.rlv[s.c1] = h.rlv;
    if (kp.eddyModel) b.rev[s.c1] = h.rev;
    b.w[s.c1 + 4 * nb] = ovgm1 * h.p + 0.5 * h.rho * (h.u * h.u + h.v * h.v + h.w * h.w);
    if (!second) return;
    const double factor = 0.5;
    const double r0 = fmax(factor * h.rho, 2.0 * h.rho - i2.rho);
    const double u0 = 2.0 * h.u - i2.u, v0 = 2.0 * h.v - i2.v, w0 = 2.0 * h.w - i2.w;
    const double p0 = fmax(factor * h.p, 2.0 * h.p - i2.p);
    b.w[s.c0] = r0; b.w[s.c0 + nb] = u0; b.w[s.c0 + 2 * nb] = v0; b.w[s.c0 + 3 * nb] = w0;
    b.p[s.c0] = p0;
    if (kp.viscous) b.rlv[s.c0] = h.rlv;
    if (kp.eddyModel) b.rev[s.c0] = h.rev;
    b.w[s.c0 + 4 * nb] = ovgm1 * p0 + 0.5 * r0 * (u0 * u0 + v0 * v0 + w0 * w0);
}

// symmetry: layer 1 mirrors slab 2, layer 0 mirrors slab 3 (two separate passes in the reference)
__device__ __forceinline__ void bcc_symm(const BlkView& b, const BcFaceDev& f, const BcSlab& s, const KParams& kp, int second)
{
    const long nb = b.nbox;
    const long ch = second ? s.c0 : s.c1, cd = second ? s.c3 : s.c2;
    const double nx = f.norm[s.f], ny = f.norm[s.f + s.fn], nz = f.norm[s.f + 2 * s.fn];
    const double u = b.w[cd + nb], v = b.w[cd + 2 * nb], w = b.w[cd + 3 * nb];
    const double vn = 2.0 * (u * nx + v * ny + w * nz);
    b.w[ch] = b.w[cd];
    b.w[ch + nb] = u - vn * nx;
    b.w[ch + 2 * nb] = v - vn * ny;
    b.w[ch + 3 * nb] = w - vn * nz;
    b.w[ch + 4 * nb] = b.w[cd + 4 * nb];
    b.gamma[ch] = b.gamma[cd];
    b.p[ch] = b.p[cd];
    if (kp.viscous) b.rlv[ch] = b.rlv[cd];
    if (kp.eddyModel) b.rev[ch] = b.rev[cd];
}
__global__ __launch_bounds__(256) void k_bc_symm(BC_ARGS, KParams kp, int second)
{
    BC_PROLOGUE
    BcSlab s;
    if (!bc_slab(b, f, s)) return;
    bcc_symm(b, f, s, kp, second);
}

// polar symmetry: the mirror direction is the diagonal of the face cell (degenerate "axis" faces), xx(i+1,j+1) - xx(i,j)
// of the face's node plane (BCRoutines.F90:370-379, setBCPointers utils.F90:1103-1133)
__device__ __forceinline__ void bcc_symm_polar(const BlkView& b, const BcFaceDev& f, const BcSlab& s, const KParams& kp, int second)
{
    const long nb = b.nbox;
    const int isize = f.icEnd - f.icBeg + 1;
    const int i = f.icBeg + (int)(s.f % isize), j = f.jcBeg + (int)(s.f / isize);
    long nA, nB;
    switch (f.faceID) {
    case ADFLOW_IMIN: case ADFLOW_IMAX: { const int P = (f.faceID == ADFLOW_IMIN) ? 1 : b.il; nA = b.idx(P, i - 1, j - 1); nB = b.idx(P, i, j); break; }
    case ADFLOW_JMIN: case ADFLOW_JMAX: { const int P = (f.faceID == ADFLOW_JMIN) ? 1 : b.jl; nA = b.idx(i - 1, P, j - 1); nB = b.idx(i, P, j); break; }
    default: { const int P = (f.faceID == ADFLOW_KMIN) ? 1 : b.kl; nA = b.idx(i - 1, j - 1, P); nB = b.idx(i, j, P); }
    }
    double nx = b.x[nB] - b.x[nA], ny = b.x[nB + nb] - b.x[nA + nb], nz = b.x[nB + 2 * nb] - b.x[nA + 2 * nb];
    const double tmp = 1.0 / sqrt(nx * nx + ny * ny + nz * nz);
    nx *= tmp; ny *= tmp; nz *= tmp;
    const long ch = second ? s.c0 : s.c1, cd = second ? s.c3 : s.c2;
    const double u = b.w[cd + nb], v = b.w[cd + 2 * nb], w = b.w[cd + 3 * nb];
    const double t2 = 2.0 * (u * nx + v * ny + w * nz);
    b.w[ch] = b.w[cd];
    b.w[ch + nb] = t2 * nx - u;
    b.w[ch + 2 * nb] = t2 * ny - v;
    b.w[ch + 3 * nb] = t2 * nz - w;
    b.w[ch + 4 * nb] = b.w[cd + 4 * nb];
    b.p[ch] = b.p[cd];
    if (kp.viscous) b.rlv[ch] = b.rlv[cd];
    if (kp.eddyModel) b.rev[ch] = b.rev[cd];
}
__global__ __launch_bounds__(256) void k_bc_symm_polar(BC_ARGS, KParams kp, int second)
{
    BC_PROLOGUE
    BcSlab s;
    if (!bc_slab(b, f, s)) return;
    bcc_symm_polar(b, f, s, kp, second);
}

// subsonic outflow / outflow mass bleed: static pressure prescribed, entropy, tangential velocity and the outgoing
// acoustic Riemann variable extrapolated
__device__ __forceinline__ void bcc_subsonic_outflow(const BlkView& b, const BcFaceDev& f, const BcSlab& s, const KParams& kp, int second)
{
    const long nb = b.nbox;
    const double pExit = f.ps[s.f];
    const double nx = f.norm[s.f], ny = f.norm[s.f + s.fn], nz = f.norm[s.f + 2 * s.fn];
    const double gam2 = b.gamma[s.c2];
    const double ovg = 1.0 / gam2, ovgm1 = 1.0 / (gam2 - 1.0);
    const double pInt = b.p[s.c2];
    const double r = 1.0 / b.w[s.c2];
    double a = sqrt(gam2 * pInt * r);
    const double ue = b.w[s.c2 + nb], ve = b.w[s.c2 + 2 * nb], we = b.w[s.c2 + 3 * nb];
    const double qne = ue * nx + ve * ny + we * nz;
    const double ss = pInt * pow(r, gam2);
    const double ac = qne + 2.0 * a * ovgm1;
    const double rho1 = pow(pExit / ss, ovg);
    b.w[s.c1] = rho1;
    b.p[s.c1] = pExit;
    a = sqrt(gam2 * pExit / rho1);
    const double qnh = ac - 2.0 * a * ovgm1;
    b.w[s.c1 + nb] = ue + (qnh - qne) * nx;
    b.w[s.c1 + 2 * nb] = ve + (qnh - qne) * ny;
    b.w[s.c1 + 3 * nb] = we + (qnh - qne) * nz;
    if (kp.viscous) b.rlv[s.c1] = b.rlv[s.c2];
    if (kp.eddyModel) b.rev[s.c1] = b.rev[s.c2];
    bc_etot(b, kp, s.c1);
    if (second) bc_second_halo(b, kp, s);
}
__global__ __launch_bounds__(256) void k_bc_subsonic_outflow(BC_ARGS, KParams kp, int second)
{
    BC_PROLOGUE
    BcSlab s;
    if (!bc_slab(b, f, s)) return;
    bcc_subsonic_outflow(b, f, s, kp, second);
}

// subsonic inflow: total conditions + flow direction, or density + velocity prescribed; the outgoing acoustic
// Riemann variable comes from the interior
__device__ __forceinline__ void bcc_subsonic_inflow(const BlkView& b, const BcFaceDev& f, const BcSlab& s, const KParams& kp, int second, int hScalingInlet)
{
    const long nb = b.nbox;
    const double nx = f.norm[s.f], ny = f.norm[s.f + s.fn], nz = f.norm[s.f + 2 * s.fn];
    const double gam2 = b.gamma[s.c2];
    const double gm1 = gam2 - 1.0, ovgm1 = 1.0 / gm1;
    const double r = 1.0 / b.w[s.c2];
    const double u2 = b.w[s.c2 + nb], v2 = b.w[s.c2 + 2 * nb], w2 = b.w[s.c2 + 3 * nb];
    double a2 = gam2 * b.p[s.c2] * r;
    double beta = u2 * nx + v2 * ny + w2 * nz + 2.0 * ovgm1 * sqrt(a2);
    if (f.inletTreatment == ADFLOW_INLET_TOTAL_CONDITIONS) {
        const double govgm1 = kp.gammaConstant / (kp.gammaConstant - 1.0);
        const double ptot = f.pt[s.f], ttot = f.tt[s.f], htot = f.ht[s.f];
        const double ssx = f.fdx[s.f], ssy = f.fdy[s.f], ssz = f.fdz[s.f];
        const double h2 = r * (b.w[s.c2 + 4 * nb] + b.p[s.c2]);
        double scaleFact = 1.0;
        if (hScalingInlet) scaleFact = sqrt(htot / h2);
        beta = beta * scaleFact;
        double q2 = u2 * u2 + v2 * v2 + w2 * w2;
        const double a2tot = gm1 * (htot - h2 + 0.5 * q2) + a2;
        const double alpha = nx * ssx + ny * ssy + nz * ssz;
        const double aa2 = 0.5 * gm1 * alpha * alpha + 1.0;
        const double bb = -gm1 * alpha * beta;
        const double cc = 0.5 * gm1 * beta * beta - 2.0 * ovgm1 * a2tot;
        double dd = bb * bb - 4.0 * aa2 * cc;
        dd = sqrt(fmax(0.0, dd));
        double q = (-bb + dd) / (2.0 * aa2);
        q = fmax(0.0, q);
        q2 = q * q;
        a2 = a2tot - 0.5 * gm1 * q2;
        double m2 = q2 / a2;
        m2 = fmin(1.0, m2);
        q2 = m2 * a2;
        q = sqrt(q2);
        a2 = a2tot - 0.5 * gm1 * q2;
        b.w[s.c1 + nb] = q * ssx;
        b.w[s.c1 + 2 * nb] = q * ssy;
        b.w[s.c1 + 3 * nb] = q * ssz;
        const double ts = a2 / (gam2 * kp.RGas);
        const double ratio = pow(ts / ttot, govgm1);
        b.p[s.c1] = ptot * ratio;
        b.w[s.c1] = (ptot * ratio) / (kp.RGas * ts);
    } else {
        const double rho = f.rho[s.f], velx = f.vx[s.f], vely = f.vy[s.f], velz = f.vz[s.f];
        a2 = 0.5 * gm1 * (beta - velx * nx - vely * ny - velz * nz);
        a2 = fmax(0.0, a2);
        a2 = a2 * a2;
        b.p[s.c1] = rho * a2 / gam2;
        b.w[s.c1] = rho;
        b.w[s.c1 + nb] = velx;
        b.w[s.c1 + 2 * nb] = vely;
        b.w[s.c1 + 3 * nb] = velz;
    }
    if (kp.viscous) b.rlv[s.c1] = b.rlv[s.c2];
    if (kp.eddyModel) b.rev[s.c1] = b.rev[s.c2];
    bc_etot(b, kp, s.c1);
    if (second) bc_second_halo(b, kp, s);
}
__global__ __launch_bounds__(256) void k_bc_subsonic_inflow(BC_ARGS, KParams kp, int second, int hScalingInlet)
{
    BC_PROLOGUE
    BcSlab s;
    if (!bc_slab(b, f, s)) return;
    bcc_subsonic_inflow(b, f, s, kp, second, hScalingInlet);
}

// viscous walls; ISO: isothermal (wall temperature TNS_Wall)
template <bool ISO>
__device__ __forceinline__ void bcc_nswall(const BlkView& b, const BcFaceDev& f, const BcSlab& s, const KParams& kp, int second, int wallTreatment)
{
    const long nb = b.nbox;
    const double rhok = 0.0;     // correctForK = .false. (no k equation)
    BcState i2, h;
    i2.rho = b.w[s.c2]; i2.u = b.w[s.c2 + nb]; i2.v = b.w[s.c2 + 2 * nb]; i2.w = b.w[s.c2 + 3 * nb]; i2.p = b.p[s.c2];
    if (wallTreatment == ADFLOW_WALLBC_CONSTANT) {
        h.p = i2.p - 4.0 * (1.0 / 3.0) * rhok;
    } else {
        h.p = 2 * i2.p - b.p[s.c3];
        if (h.p <= 0.0) h.p = i2.p;
    }
    if (ISO) {
        const double tw = f.tns[s.f];
        const double t2 = i2.p / (kp.RGas * i2.rho);
        double t1 = 2.0 * tw - t2;
        t1 = fmax(0.5 * tw, t1);
        t1 = fmin(2.0 * tw, t1);
        h.rho = h.p / (kp.RGas * t1);
    } else {
        h.rho = i2.rho;
    }
    const double us0 = f.uslip ? f.uslip[s.f] : 0.0, us1 = f.uslip ? f.uslip[s.f + s.fn] : 0.0, us2 = f.uslip ? f.uslip[s.f + 2 * s.fn] : 0.0;
    h.u = -i2.u + 2.0 * us0; h.v = -i2.v + 2.0 * us1; h.w = -i2.w + 2.0 * us2;
    h.rlv = b.rlv[s.c2];
    h.rev = kp.eddyModel ? -b.rev[s.c2] : 0.0;
    // (the laminar viscosity of a viscous wall's halo is stored whatever kp.viscous says: the kind exists on viscous meshes only)
    bc_store_halos(b, kp, s, h, i2, second != 0);
}
template <bool ISO>
__global__ __launch_bounds__(256) void k_bc_nswall(BC_ARGS, KParams kp, int second, int wallTreatment)
{
    BC_PROLOGUE
    BcSlab s;
    if (!bc_slab(b, f, s)) return;
    bcc_nswall<ISO>(b, f, s, kp, second, wallTreatment);
}

// myDim (utils): max(x - y, 0)
__device__ __forceinline__ double bc_mydim(double x, double y) { return fmax(x - y, 0.0); }

__device__ __forceinline__ void bcc_eulerwall(const BlkView& b, const BcFaceDev& f, const BcSlab& s, const KParams& kp, int second, int wallTreatment)
{
    const long nb = b.nbox;
    double grad = 0.0;
    if (wallTreatment == ADFLOW_WALLBC_LINEAR) grad = b.p[s.c3] - b.p[s.c2];
    const double nx = f.norm[s.f], ny = f.norm[s.f + s.fn], nz = f.norm[s.f + 2 * s.fn];
    const double rface = f.rface ? f.rface[s.f] : 0.0;
    const double u = b.w[s.c2 + nb], v = b.w[s.c2 + 2 * nb], w = b.w[s.c2 + 3 * nb];
    if (wallTreatment == ADFLOW_WALLBC_NORMAL_MOMENTUM) {
        // pressure gradient from the normal momentum equation (BCRoutines.F90:1123-1234): central differences of the unit
        // normal and of the interior pressure along the two directions of the face, clipped to the subface; ssi = the normal of
        // the boundary face, ssj / ssk = the face normals of the first interior cell in the generic j / k directions
        // (setBCPointers, utils.F90:1100-1137).  Blocks at rest only (the grid velocity s of the cell centre is not mirrored).
        const int isize = f.icEnd - f.icBeg + 1;
        const int a = f.icBeg + (int)(s.f % isize), q = f.jcBeg + (int)(s.f / isize);
        long sa, sb;
        const adf_real8 *Sn, *Sa, *Sb;
        switch (f.faceID) {
        case ADFLOW_IMIN: case ADFLOW_IMAX: sa = b.ldi; sb = b.ldk; Sn = b.sI; Sa = b.sJ; Sb = b.sK; break;
        case ADFLOW_JMIN: case ADFLOW_JMAX: sa = 1; sb = b.ldk; Sn = b.sJ; Sa = b.sI; Sb = b.sK; break;
        default: sa = 1; sb = b.ldi; Sn = b.sK; Sa = b.sI; Sb = b.sJ; break;
        }
        const long cf = (f.faceID == ADFLOW_IMIN || f.faceID == ADFLOW_JMIN || f.faceID == ADFLOW_KMIN) ? s.c1 : s.c2;   // the boundary face
        const int am1 = (a - 1 > f.icBeg) ? a - 1 : f.icBeg, ap1 = (a + 1 < f.icEnd) ? a + 1 : f.icEnd;
        const int qm1 = (q - 1 > f.jcBeg) ? q - 1 : f.jcBeg, qp1 = (q + 1 < f.jcEnd) ? q + 1 : f.jcEnd;
        const double a1 = 1.0 / (double)((ap1 - am1 > 1) ? ap1 - am1 : 1), b1 = 1.0 / (double)((qp1 - qm1 > 1) ? qp1 - qm1 : 1);
        const double sixa = 2.0 * Sn[cf], siya = 2.0 * Sn[cf + nb], siza = 2.0 * Sn[cf + 2 * nb];
        const double sjxa = Sa[s.c2 - sa] + Sa[s.c2], sjya = Sa[s.c2 - sa + nb] + Sa[s.c2 + nb],
                     sjza = Sa[s.c2 - sa + 2 * nb] + Sa[s.c2 + 2 * nb];
        const double skxa = Sb[s.c2 - sb] + Sb[s.c2], skya = Sb[s.c2 - sb + nb] + Sb[s.c2 + nb],
                     skza = Sb[s.c2 - sb + 2 * nb] + Sb[s.c2 + 2 * nb];
        const long fjp = s.f + (ap1 - a), fjm = s.f + (am1 - a), fkp = s.f + (long)(qp1 - q) * isize, fkm = s.f + (long)(qm1 - q) * isize;
        const double rxj = a1 * (f.norm[fjp] - f.norm[fjm]), ryj = a1 * (f.norm[fjp + s.fn] - f.norm[fjm + s.fn]),
                     rzj = a1 * (f.norm[fjp + 2 * s.fn] - f.norm[fjm + 2 * s.fn]);
        const double dpj = a1 * (b.p[s.c2 + (ap1 - a) * sa] - b.p[s.c2 + (am1 - a) * sa]);
        const double rxk = b1 * (f.norm[fkp] - f.norm[fkm]), ryk = b1 * (f.norm[fkp + s.fn] - f.norm[fkm + s.fn]),
                     rzk = b1 * (f.norm[fkp + 2 * s.fn] - f.norm[fkm + 2 * s.fn]);
        const double dpk = b1 * (b.p[s.c2 + (qp1 - q) * sb] - b.p[s.c2 + (qm1 - q) * sb]);
        const double ri = nx * sixa + ny * siya + nz * siza;
        const double rj = nx * sjxa + ny * sjya + nz * sjza;
        const double rk = nx * skxa + ny * skya + nz * skza;
        const double qj = u * sjxa + v * sjya + w * sjza;
        const double qk = u * skxa + v * skya + w * skza;
        grad = ((qj * (u * rxj + v * ryj + w * rzj) + qk * (u * rxk + v * ryk + w * rzk)) * b.w[s.c2] - rj * dpj - rk * dpk) / ri;
    }
    b.p[s.c1] = bc_mydim(b.p[s.c2], grad);
    const double vn = 2.0 * (rface - u * nx - v * ny - w * nz);
    b.w[s.c1] = b.w[s.c2];
    b.w[s.c1 + nb] = u + vn * nx;
    b.w[s.c1 + 2 * nb] = v + vn * ny;
    b.w[s.c1 + 3 * nb] = w + vn * nz;
    if (kp.viscous) b.rlv[s.c1] = b.rlv[s.c2];
    if (kp.eddyModel) b.rev[s.c1] = b.rev[s.c2];
    bc_etot(b, kp, s.c1);
    if (second) bc_second_halo(b, kp, s);
}
__global__ __launch_bounds__(256) void k_bc_eulerwall(BC_ARGS, KParams kp, int second, int wallTreatment)
{
    BC_PROLOGUE
    BcSlab s;
    if (!bc_slab(b, f, s)) return;
    bcc_eulerwall(b, f, s, kp, second, wallTreatment);
}

__device__ __forceinline__ void bcc_farfield(const BlkView& b, const BcFaceDev& f, const BcSlab& s, const KParams& kp, int second)
{
    const long nb = b.nbox;
    const double gm1 = kp.gammaInf - 1.0;
    const double ovgm1 = 1.0 / gm1;
    const double r0 = 1.0 / kp.wInf[0];
    const double u0 = kp.wInf[1], v0 = kp.wInf[2], w0 = kp.wInf[3];
    const double c0 = sqrt(kp.gammaInf * kp.pInfCorr * r0);
    const double s0 = pow(kp.wInf[0], kp.gammaInf) / kp.pInfCorr;
    const double nx = f.norm[s.f], ny = f.norm[s.f + s.fn], nz = f.norm[s.f + 2 * s.fn];
    const double rface = f.rface ? f.rface[s.f] : 0.0;
    const double qn0 = u0 * nx + v0 * ny + w0 * nz;
    const double vn0 = qn0 - rface;
    const double rho2 = b.w[s.c2], gam2 = b.gamma[s.c2], p2 = b.p[s.c2];
    const double re = 1.0 / rho2;
    const double ue = b.w[s.c2 + nb], ve = b.w[s.c2 + 2 * nb], we = b.w[s.c2 + 3 * nb];
    const double qne = ue * nx + ve * ny + we * nz;
    const double ce = sqrt(gam2 * p2 * re);
    double ac1, ac2;
    if (vn0 > -c0) ac1 = qne + 2.0 * ovgm1 * ce;      // outflow or subsonic inflow
    else ac1 = qn0 + 2.0 * ovgm1 * c0;                // supersonic inflow
    if (vn0 > c0) ac2 = qne - 2.0 * ovgm1 * ce;       // supersonic outflow
    else ac2 = qn0 - 2.0 * ovgm1 * c0;                // inflow or subsonic outflow
    const double qnf = 0.5 * (ac1 + ac2);
    const double cf = 0.25 * (ac1 - ac2) * gm1;
    double uf, vf, wf, sf;
    if (vn0 > 0.0) {                                  // outflow
        uf = ue + (qnf - qne) * nx;
        vf = ve + (qnf - qne) * ny;
        wf = we + (qnf - qne) * nz;
        sf = pow(rho2, gam2) / p2;
    } else {                                          // inflow
        uf = u0 + (qnf - qn0) * nx;
        vf = v0 + (qnf - qn0) * ny;
        wf = w0 + (qnf - qn0) * nz;
        sf = s0;
    }
    const double cc = cf * cf / gam2;
    BcState h, i2;
    h.rho = pow(sf * cc, ovgm1);
    h.u = uf; h.v = vf; h.w = wf;
    h.p = h.rho * cc;
    h.rlv = kp.viscous ? b.rlv[s.c2] : 0.0;
    h.rev = kp.eddyModel ? b.rev[s.c2] : 0.0;
    i2.rho = rho2; i2.u = ue; i2.v = ve; i2.w = we; i2.p = p2;
    bc_store_halos(b, kp, s, h, i2, second != 0);
}
__global__ __launch_bounds__(256) void k_bc_farfield(BC_ARGS, KParams kp, int second)
{
    BC_PROLOGUE
    BcSlab s;
    if (!bc_slab(b, f, s)) return;
    bcc_farfield(b, f, s, kp, second);
}

// extrap / supersonic outflow: fw2, fw3 = weights of slab 2 and 3
__device__ __forceinline__ void bcc_extrap(const BlkView& b, const BcFaceDev& f, const BcSlab& s, const KParams& kp, int second, int outflowTreatment)
{
    // extrap: linear; supersonic outflow: constant or linear (BCRoutines.F90:1512-1531)
    double fw2 = 2.0, fw3 = -1.0;
    if (f.type == ADFLOW_BC_SUPERSONIC_OUTFLOW && outflowTreatment == 1) { fw2 = 1.0; fw3 = 0.0; }
    const long nb = b.nbox;
    const double factor = 0.5;
    BcState h, i2;
    i2.rho = b.w[s.c2]; i2.u = b.w[s.c2 + nb]; i2.v = b.w[s.c2 + 2 * nb]; i2.w = b.w[s.c2 + 3 * nb]; i2.p = b.p[s.c2];
    h.rho = fmax(factor * i2.rho, fw2 * i2.rho + fw3 * b.w[s.c3]);
    h.u = fw2 * i2.u + fw3 * b.w[s.c3 + nb];
    h.v = fw2 * i2.v + fw3 * b.w[s.c3 + 2 * nb];
    h.w = fw2 * i2.w + fw3 * b.w[s.c3 + 3 * nb];
    h.p = fmax(factor * i2.p, fw2 * i2.p + fw3 * b.p[s.c3]);
    h.rlv = kp.viscous ? b.rlv[s.c2] : 0.0;
    h.rev = kp.eddyModel ? b.rev[s.c2] : 0.0;
    bc_store_halos(b, kp, s, h, i2, second != 0);
}
__global__ __launch_bounds__(256) void k_bc_extrap(BC_ARGS, KParams kp, int second, int outflowTreatment)
{
    BC_PROLOGUE
    BcSlab s;
    if (!bc_slab(b, f, s)) return;
    bcc_extrap(b, f, s, kp, second, outflowTreatment);
}

// supersonic inflow (BCRoutines.F90:1411-1477): both halo layers take the prescribed state
__device__ __forceinline__ void bcc_supersonic_inflow(const BlkView& b, const BcFaceDev& f, const BcSlab& s, const KParams& kp, int second)
{
    const long nb = b.nbox;
    b.w[s.c1] = f.rho[s.f];
    b.w[s.c1 + nb] = f.vx[s.f];
    b.w[s.c1 + 2 * nb] = f.vy[s.f];
    b.w[s.c1 + 3 * nb] = f.vz[s.f];
    b.p[s.c1] = f.ps[s.f];
    if (kp.viscous) b.rlv[s.c1] = b.rlv[s.c2];
    if (kp.eddyModel) b.rev[s.c1] = b.rev[s.c2];
    bc_etot(b, kp, s.c1);
    if (second) {
        b.w[s.c0] = b.w[s.c1];
        b.w[s.c0 + nb] = b.w[s.c1 + nb];
        b.w[s.c0 + 2 * nb] = b.w[s.c1 + 2 * nb];
        b.w[s.c0 + 3 * nb] = b.w[s.c1 + 3 * nb];
        b.p[s.c0] = b.p[s.c1];
        if (kp.viscous) b.rlv[s.c0] = b.rlv[s.c1];
        if (kp.eddyModel) b.rev[s.c0] = b.rev[s.c1];
        bc_etot(b, kp, s.c0);
    }
}
__global__ __launch_bounds__(256) void k_bc_supersonic_inflow(BC_ARGS, KParams kp, int second)
{
    BC_PROLOGUE
    BcSlab s;
    if (!bc_slab(b, f, s)) return;
    bcc_supersonic_inflow(b, f, s, kp, second);
}

static dim3 bc_grid(const BcPhase& ph) { return dim3((unsigned)((ph.maxCells + 255) / 256), (unsigned)ph.count, 1); }

// applyAllBC for every block of the level: `flow` = launch plan in the reference's order (symm, symm 2nd halo pass,
// adiabatic walls, isothermal walls, farfield, extrap / supersonic outflow, Euler wall, supersonic inflow; inside a
// kind one launch per ordinal of the subface within its block)
void launch_apply_all_bc(const BlkView* tab, const BcEntry* ent, const int* order, const std::vector<BcPhase>& flow,
                         const KParams& kp, int second, int eulerWallTreatment, int viscWallTreatment, int outflowTreatment,
                         int hScalingInlet, hipStream_t s)
{
    const dim3 blk(256, 1, 1);
    // coarse levels force the constant-pressure wall treatment (BCRoutines.F90:552-553, 1098-1099)
    if (!kp.fineGrid) { eulerWallTreatment = ADFLOW_WALLBC_CONSTANT; viscWallTreatment = ADFLOW_WALLBC_CONSTANT; }
    for (const BcPhase& ph : flow) {
        const int* o = order + ph.first;
        switch (ph.kind) {
        case BCP_SYMM1: hipLaunchKernelGGL(k_bc_symm, bc_grid(ph), blk, 0, s, tab, ent, o, kp, 0); break;
        case BCP_SYMM2: if (second) hipLaunchKernelGGL(k_bc_symm, bc_grid(ph), blk, 0, s, tab, ent, o, kp, 1); break;
        case BCP_WALL_ADIABATIC: hipLaunchKernelGGL((k_bc_nswall<false>), bc_grid(ph), blk, 0, s, tab, ent, o, kp, second, viscWallTreatment); break;
        case BCP_WALL_ISOTHERMAL: hipLaunchKernelGGL((k_bc_nswall<true>), bc_grid(ph), blk, 0, s, tab, ent, o, kp, second, viscWallTreatment); break;
        case BCP_FARFIELD: hipLaunchKernelGGL(k_bc_farfield, bc_grid(ph), blk, 0, s, tab, ent, o, kp, second); break;
        case BCP_EXTRAP: hipLaunchKernelGGL(k_bc_extrap, bc_grid(ph), blk, 0, s, tab, ent, o, kp, second, outflowTreatment); break;
        case BCP_EULERWALL: hipLaunchKernelGGL(k_bc_eulerwall, bc_grid(ph), blk, 0, s, tab, ent, o, kp, second, eulerWallTreatment); break;
        case BCP_SUPERSONIC_INFLOW: hipLaunchKernelGGL(k_bc_supersonic_inflow, bc_grid(ph), blk, 0, s, tab, ent, o, kp, second); break;
        case BCP_SYMMPOLAR1: hipLaunchKernelGGL(k_bc_symm_polar, bc_grid(ph), blk, 0, s, tab, ent, o, kp, 0); break;
        case BCP_SYMMPOLAR2: if (second) hipLaunchKernelGGL(k_bc_symm_polar, bc_grid(ph), blk, 0, s, tab, ent, o, kp, 1); break;
        case BCP_SUBSONIC_OUTFLOW: hipLaunchKernelGGL(k_bc_subsonic_outflow, bc_grid(ph), blk, 0, s, tab, ent, o, kp, second); break;
        case BCP_SUBSONIC_INFLOW: hipLaunchKernelGGL(k_bc_subsonic_inflow, bc_grid(ph), blk, 0, s, tab, ent, o, kp, second, hScalingInlet); break;
        default: break;
        }
    }
}

// ---------------------------------------------------------------------------
// Turbulence boundary conditions of Spalart-Allmaras (src/turbulence/turbBCRoutines.F90)
//   bcTurbTreatment           :662-798  bmt/bvt of every face: zero, then per subface
//     bcTurbWall (SA)         :799-870  bmt = 1      (nuTilde_halo = -nuTilde_interior)
//     bcTurbSymm / outflow    :614-660, 564-613  bmt = -1 (copy)
//     bcTurbInflow            :460-515  bvt = 2 turbInlet, bmt = 1 (face value = turbInlet)
//     bcTurbFarfield          :373-459  outflow: bmt = -1, inflow: bvt = wInf(itu1)
//   applyAllTurbBCThisBlock   :49-236   halo = bvt - bmt * interior; eddy viscosity
//                                       -rev (walls) / +rev (others); turb2ndHalo copies
// ---------------------------------------------------------------------------
__device__ __forceinline__ long bc_face_entry(const BlkView& b, const BcFaceDev& f, long t, int* fidx)
{
    const int isize = f.icEnd - f.icBeg + 1;
    const int i = f.icBeg + (int)(t % isize), j = f.jcBeg + (int)(t / isize);
    const int A = (f.faceID <= ADFLOW_IMAX) ? b.je : b.ie;
    *fidx = f.faceID - 1;
    return (long)(i - 1) + (long)A * (j - 1);
}

__global__ __launch_bounds__(256) void k_turb_bc_zero(const BlkView* __restrict__ tab)
{
    const BlkView& b = tab[blockIdx.y + 1];
    if (b.nx == 0 || !b.bmt[0]) return;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n[3] = {(long)b.je * b.ke, (long)b.ie * b.ke, (long)b.ie * b.je};
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (t < n[d]) {
            b.bmt[2 * d][t] = 0.0; b.bmt[2 * d + 1][t] = 0.0;
            b.bvt[2 * d][t] = 0.0; b.bvt[2 * d + 1][t] = 0.0;
        }
}

// bmt / bvt of cell t of a subface (bcTurbTreatment)
__device__ __forceinline__ void bcc_turb_treatment(const BlkView& b, const BcFaceDev& f, long t, const KParams& kp)
{
    const long n = (long)(f.icEnd - f.icBeg + 1) * (f.jcEnd - f.jcBeg + 1);
    // the face arrays only cover 1..ie x 1..je (turbBCRoutines.F90:684-735): skip range cells outside
    const int isize = f.icEnd - f.icBeg + 1;
    const int i = f.icBeg + (int)(t % isize), j = f.jcBeg + (int)(t / isize);
    const int A = (f.faceID <= ADFLOW_IMAX) ? b.je : b.ie;
    const int B = (f.faceID <= ADFLOW_JMAX) ? b.ke : b.je;
    if (i < 1 || i > A || j < 1 || j > B) return;
    int fi;
    const long e = bc_face_entry(b, f, t, &fi);
    switch (f.type) {
    case ADFLOW_BC_NSWALL_ADIABATIC: case ADFLOW_BC_NSWALL_ISOTHERMAL:
        b.bmt[fi][e] = 1.0;
        break;
    case ADFLOW_BC_SYMM: case ADFLOW_BC_SYMM_POLAR: case ADFLOW_BC_EULERWALL:
        b.bmt[fi][e] = -1.0;
        break;
#ifndef ADF_AD_BUILD
    // The in- and outflow kinds (bcTurbInflow / bcTurbOutflow) stand inside `#ifndef USE_TAPENADE` in the reference
    // (turbBCRoutines.F90:743-763): its forward-mode code leaves bmt = bvt = 0 on those subfaces -- the halo takes nuTilde = 0 with
    // a zero derivative.  The dual build (kernels_ad.hip) reproduces that.
    case ADFLOW_BC_SUPERSONIC_OUTFLOW: case ADFLOW_BC_EXTRAP: case ADFLOW_BC_SUBSONIC_OUTFLOW: case ADFLOW_BC_MASSBLEED_OUTFLOW:
        b.bmt[fi][e] = -1.0;
        break;
    case ADFLOW_BC_SUPERSONIC_INFLOW: case ADFLOW_BC_SUBSONIC_INFLOW:     // bcTurbInflow (turbBCRoutines.F90:460-515)
        if (f.turbInlet) {
            b.bvt[fi][e] = 2.0 * f.turbInlet[t];
            b.bmt[fi][e] = 1.0;
        }
        break;
#endif
    case ADFLOW_BC_FARFIELD: {
        const double dot = f.norm[t] * kp.wInf[1] + f.norm[t + n] * kp.wInf[2] + f.norm[t + 2 * n] * kp.wInf[3] -
                           (f.rface ? f.rface[t] : 0.0);
        if (dot > 0.0) b.bmt[fi][e] = -1.0;
        else b.bvt[fi][e] = kp.wInf[5];
        break;
    }
    default: break;
    }
}
__global__ __launch_bounds__(256) void k_turb_bc_treatment(BC_ARGS, KParams kp)
{
    BC_PROLOGUE
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)(f.icEnd - f.icBeg + 1) * (f.jcEnd - f.jcBeg + 1);
    if (t >= n || !b.bmt[0]) return;
    bcc_turb_treatment(b, f, t, kp);
}

__device__ __forceinline__ void bcc_turb_apply(const BlkView& b, const BcFaceDev& f, const BcSlab& s, const KParams& kp, int second)
{
    int fi;
    const long e = bc_face_entry(b, f, s.f, &fi);
    const long nt = 5 * b.nbox;
    b.w[s.c1 + nt] = b.bvt[fi][e] - b.bmt[fi][e] * b.w[s.c2 + nt];
    if (kp.eddyModel) {
        const bool wall = (f.type == ADFLOW_BC_NSWALL_ADIABATIC || f.type == ADFLOW_BC_NSWALL_ISOTHERMAL);
        b.rev[s.c1] = wall ? -b.rev[s.c2] : b.rev[s.c2];
    }
    if (second) {
        b.w[s.c0 + nt] = b.w[s.c1 + nt];
        if (kp.eddyModel) b.rev[s.c0] = b.rev[s.c1];
    }
}
__global__ __launch_bounds__(256) void k_apply_turb_bc(BC_ARGS, KParams kp, int second)
{
    BC_PROLOGUE
    BcSlab s;
    if (!bc_slab(b, f, s) || !b.bmt[0]) return;
    bcc_turb_apply(b, f, s, kp, second);
}


// `ordinal`: one launch per ordinal of a subface within its block (the r-th subfaces of all blocks together)
void launch_turb_bc_treatment(const BlkView* tab, int nslots, long maxFace, const BcEntry* ent, const int* order,
                              const std::vector<BcPhase>& ordinal, const KParams& kp, hipStream_t s)
{
    hipLaunchKernelGGL(k_turb_bc_zero, dim3((unsigned)((maxFace + 255) / 256), nslots), dim3(256), 0, s, tab);
    for (const BcPhase& ph : ordinal)
        hipLaunchKernelGGL(k_turb_bc_treatment, bc_grid(ph), dim3(256), 0, s, tab, ent, order + ph.first, kp);
}

void launch_apply_turb_bc(const BlkView* tab, const BcEntry* ent, const int* order, const std::vector<BcPhase>& ordinal,
                          const KParams& kp, int second, hipStream_t s)
{
    for (const BcPhase& ph : ordinal)
        hipLaunchKernelGGL(k_apply_turb_bc, bc_grid(ph), dim3(256), 0, s, tab, ent, order + ph.first, kp, second);
}

// ---------------------------------------------------------------------------
// Multigrid pieces that touch boundary halos (src/solver/multiGrid.F90)
//   setCornerRowHalos         :1032-1357  after the restriction: first-halo cells next to the block edges take
//                                         the value of their interior neighbour so that boundary conditions that
//                                         read them find defined data.  Six loops, each reading what earlier
//                                         ones wrote: one workgroup walks them in order.
//   setCorrectionsCoarseHalos :1359-1503  corrections in the boundary halos of the coarse block before the
//                                         prolongation: symmetry mirrors them, every other kind scales them by
//                                         fact (0 for mgBoundCorr = bcDirichlet0, the only value the reference sets)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void crh_copy(const BlkView& b, const KParams& kp, long dst, long src)
{
#pragma unroll
    for (int l = 0; l < 5; ++l) b.w[dst + l * b.nbox] = b.w[src + l * b.nbox];
    b.p[dst] = b.p[src];
    if (kp.viscous) b.rlv[dst] = b.rlv[src];
    if (kp.eddyModel) b.rev[dst] = b.rev[src];
}

__global__ __launch_bounds__(256) void k_corner_row_halos(const BlkView* __restrict__ tab, KParams kp)
{
    const BlkView& b = tab[blockIdx.x + 1];     // one workgroup per block of the level
    if (b.nx == 0) return;
    const int t = threadIdx.x, nt = blockDim.x;
    const int il = b.il, jl = b.jl, kl = b.kl, ie = b.ie, je = b.je, ke = b.ke;
    const int J4[4] = {2, (3 < jl) ? 3 : jl, jl, (2 > b.ny) ? 2 : b.ny};
    const int K4[4] = {2, (3 < kl) ? 3 : kl, kl, (2 > b.nz) ? 2 : b.nz};
    const int I4[4] = {2, (3 < il) ? 3 : il, il, (2 > b.nx) ? 2 : b.nx};
    // 1: k = 2..kl, i halos at j in J4
    for (int k = 2 + t; k <= kl; k += nt)
        for (int q = 0; q < 4; ++q) { crh_copy(b, kp, b.idx(1, J4[q], k), b.idx(2, J4[q], k)); crh_copy(b, kp, b.idx(ie, J4[q], k), b.idx(il, J4[q], k)); }
    __syncthreads();
    // 2: j = 3..ny, i halos at k in K4
    for (int j = 3 + t; j <= b.ny; j += nt)
        for (int q = 0; q < 4; ++q) { crh_copy(b, kp, b.idx(1, j, K4[q]), b.idx(2, j, K4[q])); crh_copy(b, kp, b.idx(ie, j, K4[q]), b.idx(il, j, K4[q])); }
    __syncthreads();
    // 3: k = 3..nz, j halos at i in I4
    for (int k = 3 + t; k <= b.nz; k += nt)
        for (int q = 0; q < 4; ++q) { crh_copy(b, kp, b.idx(I4[q], 1, k), b.idx(I4[q], 2, k)); crh_copy(b, kp, b.idx(I4[q], je, k), b.idx(I4[q], jl, k)); }
    __syncthreads();
    // 4: i = 1..ie, j halos at k in K4
    for (int i = 1 + t; i <= ie; i += nt)
        for (int q = 0; q < 4; ++q) { crh_copy(b, kp, b.idx(i, 1, K4[q]), b.idx(i, 2, K4[q])); crh_copy(b, kp, b.idx(i, je, K4[q]), b.idx(i, jl, K4[q])); }
    __syncthreads();
    // 5: j = 1..je, k halos at i in I4
    for (int j = 1 + t; j <= je; j += nt)
        for (int q = 0; q < 4; ++q) { crh_copy(b, kp, b.idx(I4[q], j, 1), b.idx(I4[q], j, 2)); crh_copy(b, kp, b.idx(I4[q], j, ke), b.idx(I4[q], j, kl)); }
    __syncthreads();
    // 6: i = 1..ie, k halos at j in J4
    for (int i = 1 + t; i <= ie; i += nt)
        for (int q = 0; q < 4; ++q) { crh_copy(b, kp, b.idx(i, J4[q], 1), b.idx(i, J4[q], 2)); crh_copy(b, kp, b.idx(i, J4[q], ke), b.idx(i, J4[q], kl)); }
}

void launch_corner_row_halos_level(const BlkView* tab, int nslots, const KParams& kp, hipStream_t s)
{
    if (nslots <= 0) return;
    hipLaunchKernelGGL(k_corner_row_halos, dim3(nslots), dim3(256), 0, s, tab, kp);
}

// corrections live in scratch(:,:,:,0:4) = d(rho), d(u), d(v), d(w), d(p) of the coarse block (kernels_mg.hip)
__global__ __launch_bounds__(256) void k_bc_coarse_corrections(BC_ARGS, double fact)
{
    BC_PROLOGUE
    BcSlab s;
    if (!bc_slab(b, f, s)) return;
    const long nb = b.nbox;
    double* __restrict__ q = b.scratch;
    if (f.type == ADFLOW_BC_SYMM) {
        const double nx = f.norm[s.f], ny = f.norm[s.f + s.fn], nz = f.norm[s.f + 2 * s.fn];
        const double u = q[s.c2 + nb], v = q[s.c2 + 2 * nb], w = q[s.c2 + 3 * nb];
        const double vn = 2.0 * (u * nx + v * ny + w * nz);
        q[s.c1] = q[s.c2];
        q[s.c1 + nb] = u - vn * nx;
        q[s.c1 + 2 * nb] = v - vn * ny;
        q[s.c1 + 3 * nb] = w - vn * nz;
        q[s.c1 + 4 * nb] = q[s.c2 + 4 * nb];
    } else {
#pragma unroll
        for (int l = 0; l < 5; ++l) q[s.c1 + l * nb] = fact * q[s.c2 + l * nb];
    }
}

void launch_bc_coarse_corrections(const BlkView* tab, const BcEntry* ent, const int* order, const std::vector<BcPhase>& ordinal,
                                  double fact, hipStream_t s)
{
    for (const BcPhase& ph : ordinal)
        hipLaunchKernelGGL(k_bc_coarse_corrections, bc_grid(ph), dim3(256), 0, s, tab, ent, order + ph.first, fact);
}
