// Internal device data model of the MI355X residual engine (not part of the ABI).
//
// HBM layout: every per-cell, per-face and per-node array of a
// block lives in ONE uniform index box (0:ib,0:jb,0:kb), i fastest, SoA with the
// variable index slowest.  One linear offset  idx = i + j*ldi + k*ldk  addresses
// every array, and the six neighbours are idx±1, ±ldi, ±ldk.  Rows are padded to
// ldi (multiple of 16 doubles) and the box origin is shifted by PAD0 doubles so
// that the first OWNED cell of every row (i=2) starts a 128-byte line: a
// wavefront whose lane l handles cell i=2+l issues fully aligned 512-byte loads.
//   face arrays : sI(i,j,k,:) (face between cells i and i+1) stored at cell idx
//   node arrays : x(i,j,k,:), nodal gradients stored at cell idx (node = upper
//                 corner of cell (i,j,k), as in the reference, block.F90:363)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/adflow_gpu.h"

#define ADF_PAD0 14   // (PAD0 + 2) % 16 == 0

// pointer into HBM with an explicit global address space: pointers that reach a
// kernel through a table in memory are "generic" to the compiler and would be
// accessed with the slower flat_load/flat_store forms.
#ifdef HOSTSIM
#define GPTR(T) T*
#else
#define GPTR(T) T __attribute__((address_space(1)))*
#endif

// element access through a uniform base pointer and a 32-bit BYTE offset: the
// form `global_load v, v_off, s[base]` (scalar base + 32-bit vector offset) that
// needs no 64-bit address arithmetic per access.  Valid while one component of
// one block stays below 4 GiB (512 M doubles).
__device__ __forceinline__ double ldg(GPTR(const double) base, unsigned byteoff)
{
    return *(GPTR(const double))((GPTR(const char))base + byteoff);
}
__device__ __forceinline__ void stg(GPTR(double) base, unsigned byteoff, double v)
{
    *(GPTR(double))((GPTR(char))base + byteoff) = v;
}

// 1 / b and 1 / sqrt(x) for operands in the normal range: the hardware seed (v_rcp_f64 / v_rsq_f64, quarter rate) refined
// by ADF_NR Newton steps in FMA form, without the range handling, final residual correction and denormal scaling of the
// compiler's division / sqrt.  Callers multiply by the result; the flux kernels that use them are bound by FP64 issue.
// Measured on MI355X (tools/pmc_calib.bin probe, profiles/r02_a_probe.txt): raw seeds 4.6e-8 / 5.2e-8 relative, one step
// 2.1e-15 / 4.1e-15, two steps 1.1e-16 / 2.2e-16 -> ONE step is enough for the 1e-10 parity bar.
#ifndef ADF_NR
#define ADF_NR 1
#endif
#ifdef HOSTSIM
__device__ __forceinline__ double rcp_nr(double b) { return 1.0 / b; }
__device__ __forceinline__ double rsq_nr(double x) { return 1.0 / sqrt(x); }
#else
__device__ __forceinline__ double rcp_nr(double b)
{
    double x = __builtin_amdgcn_rcp(b);
#pragma unroll
    for (int it = 0; it < ADF_NR; ++it) {
        const double e = __builtin_fma(-b, x, 1.0);
        x = __builtin_fma(x, e, x);
    }
    return x;
}
__device__ __forceinline__ double rsq_nr(double x)
{
    double y = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int it = 0; it < ADF_NR; ++it) {
        const double t = x * y;
        const double e = __builtin_fma(-t, y, 1.0);      // 1 - x y^2
        y = __builtin_fma(0.5 * y, e, y);
    }
    return y;
}
#endif

// a / b for denominators in the normal range (clamped differences, densities, volumes, sound speeds): a * rcp_nr(b),
// 5 FP64 issue slots instead of the ~17 of the compiler's division (v_div_scale / v_div_fmas / v_div_fixup range handling)
__device__ __forceinline__ double fastdiv(double a, double b) { return a * rcp_nr(b); }

// sqrt(x) for x >= 0 in the normal range or exactly 0
#ifdef HOSTSIM
__device__ __forceinline__ double fastsqrt(double x) { return sqrt(x); }
__device__ __forceinline__ double fast_root6(double x) { return pow(x, 1.0 / 6.0); }
__device__ __forceinline__ double fast_exp_neg(double x) { return exp(x); }
__device__ __forceinline__ double fast_powa(double x, double a) { return pow(x, a); }
#else
__device__ __forceinline__ double fastsqrt(double x) { return x * rsq_nr(fmax(x, 1.e-300)); }
// x^(1/6) for x in the normal positive range: single-precision seed of z = x^(-1/6) (v_log_f32 / v_exp_f32), two Newton steps
// z <- z (7 - x z^6) / 6 in FP64 (quadratic: 1e-7 -> 4e-14 -> 1e-26), result x z^5.  ~25 issue slots instead of ~150 of pow().
// The argument is first scaled into [2^-6, 2^6) by a power of 2^6 taken from its exponent (x = x' 2^(6k), x^(1/6) = x'^(1/6) 2^k), so that
// the single-precision seed never under- / overflows: x reaches 1e-100 when the SA variable is negative and rr is far below zero (the
// reference bounds rr only from above, sa.F90).  x = 0 (gg^6 overflowed) gives 0 as the reference's power does.
__device__ __forceinline__ double fast_root6(double x)
{
    const int e = __builtin_amdgcn_frexp_exp(x);
    const int k = (e >= 0 ? e : e - 5) / 6;                 // floor(e / 6)
    const double xs = __builtin_ldexp(x, -6 * k);
    double z = (double)__builtin_amdgcn_exp2f(-(1.0f / 6.0f) * __builtin_amdgcn_logf((float)xs));
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double z2 = z * z, z6 = z2 * z2 * z2;
        z = z * __builtin_fma(-xs, z6, 7.0) * (1.0 / 6.0);
    }
    const double z2 = z * z;
    const double r = __builtin_ldexp(xs * (z2 * z2 * z), k);
    return (x == 0.0) ? 0.0 : r;
}
// x^a for x in the normal positive range and moderate a (the directional scaling of the spectral radii, adis = 0.67 by default):
// x = m 2^e with m in [0.707, 1.414); ln m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.172, odd series to s^19 (truncation
// 1e-16); 2^(a log2 x) = 2^k exp(f ln 2) with the degree-13 Taylor polynomial on |f ln 2| <= 0.347 (4e-18).  ~50 issue slots
// instead of the ~200 of log() + exp().
__device__ __forceinline__ double fast_powa(double x, double a)
{
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);          // [0.5, 1)
    if (m < 0.70710678118654752) { m *= 2.0; e -= 1; }
    const double s = (m - 1.0) * rcp_nr(m + 1.0);
    const double s2 = s * s;
    double p = 1.0 / 19.0;
    p = __builtin_fma(p, s2, 1.0 / 17.0);
    p = __builtin_fma(p, s2, 1.0 / 15.0);
    p = __builtin_fma(p, s2, 1.0 / 13.0);
    p = __builtin_fma(p, s2, 1.0 / 11.0);
    p = __builtin_fma(p, s2, 1.0 / 9.0);
    p = __builtin_fma(p, s2, 1.0 / 7.0);
    p = __builtin_fma(p, s2, 1.0 / 5.0);
    p = __builtin_fma(p, s2, 1.0 / 3.0);
    p = __builtin_fma(p, s2, 1.0);
    const double log2x = __builtin_fma(2.0 * s * p, 1.4426950408889634, (double)e);
    const double t = a * log2x;
    const double k = __builtin_rint(t);
    const double r = (t - k) * 6.93147180559945309417e-01;
    double q = 1.0 / 6227020800.0;
    q = __builtin_fma(q, r, 1.0 / 479001600.0);
    q = __builtin_fma(q, r, 1.0 / 39916800.0);
    q = __builtin_fma(q, r, 1.0 / 3628800.0);
    q = __builtin_fma(q, r, 1.0 / 362880.0);
    q = __builtin_fma(q, r, 1.0 / 40320.0);
    q = __builtin_fma(q, r, 1.0 / 5040.0);
    q = __builtin_fma(q, r, 1.0 / 720.0);
    q = __builtin_fma(q, r, 1.0 / 120.0);
    q = __builtin_fma(q, r, 1.0 / 24.0);
    q = __builtin_fma(q, r, 1.0 / 6.0);
    q = __builtin_fma(q, r, 0.5);
    q = __builtin_fma(q, r, 1.0);
    q = __builtin_fma(q, r, 1.0);
    return __builtin_ldexp(q, (int)k);
}
// exp(x) for x <= 0 (the ft2 term of Spalart-Allmaras): 0 below -700, else 2^k * P(r) with k = round(x log2 e), r = x - k ln 2
// in two pieces, Taylor polynomial of degree 13 on |r| <= ln2 / 2 (truncation 1e-17), ldexp.  ~25 slots instead of ~60.
__device__ __forceinline__ double fast_exp_neg(double x)
{
    if (x < -700.0) return 0.0;
    const double k = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);
}
#endif

// binary exponent of a finite x > 0 (frexp: x = f 2^e, 0.5 <= f < 1)
#ifdef HOSTSIM
__device__ __forceinline__ int exponent_of(double x) { int e; (void)frexp(x, &e); return e; }
#else
__device__ __forceinline__ int exponent_of(double x) { return __builtin_amdgcn_frexp_exp(x); }
#endif

// a b + c in one rounding (the dual form, kernels_ad.hip, is the plain product and sum)
__device__ __forceinline__ double adf_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// high 32 bits of a double (the exponent test of the limiter's clamp, kernels_roe_march.hip; kernels_ad.hip holds the dual form)
__device__ __forceinline__ int adf_hiword(double x) { return __double2hiint(x); }

// a value that is the same in every lane of the wavefront, moved to a scalar register so that branches on it are scalar branches
#ifdef HOSTSIM
__device__ __forceinline__ int wave_uniform(int v) { return v; }
#else
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif

// value of the neighbouring lane of the 64-wide wavefront: lane_up1 = lane-1
// (__shfl_up by 1), lane_dn1 = lane+1.  On gfx950 these are single DPP moves
// (v_mov_b32_dpp wave_shr:1 / wave_shl:1) per 32-bit half instead of a
// ds_bpermute through the LDS crossbar.  Lane 0 / 63 receive garbage.
#ifdef HOSTSIM
__device__ __forceinline__ double lane_up1(double v) { return __shfl_up(v, 1); }
__device__ __forceinline__ double lane_dn1(double v) { return __shfl_down(v, 1); }
__device__ __forceinline__ int lane_up1(int v) { return __shfl_up(v, 1); }
#else
__device__ __forceinline__ int lane_up1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int lane_dn1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ double lane_up1(double v)
{
    return __hiloint2double(lane_up1(__double2hiint(v)), lane_up1(__double2loint(v)));
}
__device__ __forceinline__ double lane_dn1(double v)
{
    return __hiloint2double(lane_dn1(__double2hiint(v)), lane_dn1(__double2loint(v)));
}




#endif

typedef double adf_real8;      // a double that stays one where `double` is re-defined (kernels_ad.hip)
#define ADF_BLKVIEW BlkView
#define ADF_GEOM double
#include "blkview_def.h"
#undef ADF_GEOM
#undef ADF_BLKVIEW

// one boundary subface on the device (adflow_bc_subface with device copies of the BCData members)
struct BcFaceDev {
    int type, faceID, icBeg, icEnd, jcBeg, jcEnd;
    const double *norm, *rface, *uslip, *tns, *rho, *vx, *vy, *vz, *ps;
    const double *pt, *tt, *ht, *fdx, *fdy, *fdz;   // subsonic inflow with total conditions
    const double* turbInlet;                       // prescribed turbulence variable of inflow subfaces
    int inletTreatment, pad;
    double symNorm[3];     // BCData%symNorm of a symmetry plane (xhalo_block)
    double* tauq;          // viscous subfaces: viscSubface%tau(:,:,1:6), %q(:,:,1:3) over the OWNED face cells, component-major
};

// owned face cells of a subface (the node range inBeg+1..inEnd of viscSubfaceInfo, preprocessingAPI.F90:2520-2524)
__host__ __device__ inline void bc_owned_range(int faceID, int icBeg, int icEnd, int jcBeg, int jcEnd, int il, int jl, int kl, int r[4])
{
    const int amax = (faceID <= 2) ? jl : il, bmax = (faceID <= 4) ? kl : jl;
    r[0] = icBeg > 2 ? icBeg : 2;
    r[1] = icEnd < amax ? icEnd : amax;
    r[2] = jcBeg > 2 ? jcBeg : 2;
    r[3] = jcEnd < bmax ? jcEnd : bmax;
}

struct BcEntry { int slot, pad; BcFaceDev f; };     // one subface of the block in table slot `slot`
enum { BCP_SYMM1, BCP_SYMM2, BCP_WALL_ADIABATIC, BCP_WALL_ISOTHERMAL, BCP_FARFIELD, BCP_EXTRAP, BCP_EULERWALL,
       BCP_SUPERSONIC_INFLOW, BCP_SYMMPOLAR1, BCP_SYMMPOLAR2, BCP_SUBSONIC_OUTFLOW, BCP_SUBSONIC_INFLOW, BCP_ORDINAL };
struct BcPhase { int kind, first, count; long maxCells; };   // one launch: entries order[first .. first+count)

// porosity codes after the +1 shift used in `flags`
#define ADF_POR_NOFLUX 0
#define ADF_POR_BOUND 1
#define ADF_POR_NORMAL 2
__device__ __forceinline__ int flg_porI(uint8_t f) { return f & 3; }
__device__ __forceinline__ int flg_porJ(uint8_t f) { return (f >> 2) & 3; }
__device__ __forceinline__ int flg_porK(uint8_t f) { return (f >> 4) & 3; }
__device__ __forceinline__ double flg_blank(uint8_t f) { return (f & 64) ? 1.0 : 0.0; }

// per-call scalar parameters derived from adflow_opts on the host
struct KParams {
    int equations, spaceDiscr, limiter, orderTurb, turbProd;
    int viscous, eddyModel, dirScaling, useQCR, useRotationSA, useft2SA;
    int fineGrid;          // currentLevel == groundLevel
    int groundLevelIsOne;  // groundLevel == 1 (second-order turbulence advection only then)
    int doScaling;         // dirScaling && currentLevel <= groundLevel
    int onlyRadii;
    int coarseInit;        // initres: dw = wr instead of 0
    int updateEddy;        // currentLevel <= groundLevel: recompute rev in the stage update
    int fwMode;            // 0: fw not persistent (rFil==1, sfil==0, no store)  1: persistent fw
    int storeIntermed;     // store dtl / radii
    int dissApprox;        // lumped dissipation with the frozen sensor in b.ss (inviscidDissFlux*Approx)
    int viscFirst;         // the viscous march runs BEFORE the Roe march: it writes its flux sums into dw(2:5), the Roe march adds them and completes dw
    int radiiInMarch;      // the Euler march forms the spectral radii itself (no k_time_step pass in front)
    int metricFromX;       // marching kernels re-form the face normals from the node coordinates (as blocketteResCore, blockette.F90:854-960)
    int lumpedDiss;        // inputDiscretization::lumpedDiss (preconditioner assembly): first-order Roe upwind (fluxes.F90:1536)
    double sigma;
    double rFil, sfil;
    double vis2, vis4, vis2Coarse, adis, acousticScaleFactor, kappaCoef;
    double gammaConstant, gammaInf, pInf, pInfCorr, rhoInf, RGas, muRef, TRef, timeRef;
    double prandtl, prandtlTurb, SSuthDim, muSuthDim, TSuthDim;
    double sa_k, sa_cb1, sa_cb2, sa_cb3, sa_cv1, sa_cw1, sa_cw2, sa_cw3, sa_ct3, sa_ct4, sa_crot;
    double sa_qqFactor;    // 1 + (1-alfaTurb)/alfaTurb for implicit relaxation, else 1
    double sa_updFactor;   // alfaTurb for explicit relaxation, else 1
    double cfl, cflLimit, smoop, fcoll, turbResScale;
    double wInf[10];
    // matrix-free matvec (adflow_gpu_nk_residual_dev): the kernels that complete dw also write setRVec's entries dw / volRef
    // (NKSolvers.F90:1262-1376) of the PETSc-ordered residual vector (block offset BlkView::vecOff); NULL otherwise
    double* rvec;
    double rvecTurbScale;
    // the turbulence entry of rvec is written by the Roe march from the dw(itu1) the SA march left (both run in one queue, SA first):
    // the six entries of a cell go out together -- 48 contiguous bytes per cell instead of 40 + 8 from two kernels
    int rvecTurbFromDw;
    // Jacobian assembly (adflow_gpu_fd_jacobian on the marching kernels of the preconditioner matrix): the kernels that complete the
    // residual write the dense snapshot of the coloured evaluation -- resScale(dw), or its derivative part in forward mode -- INSTEAD
    // of dw (k_fd_snap / k_ad_snap read dw back and wrote the same numbers); NULL otherwise.  snapTab: per block slot the snapshot
    // array; component m of colour snapCol at ((snapCol snapN + m) nbox); state variable l is component l - snapL0
    const struct SnapSlot* snapTab;
    int snapCol, snapL0, snapN;
    double snapTurbScale;
};
struct SnapSlot { double* snap; };
// plain build: the scaled residual itself (k_fd_scatter forms the finite difference; kernels_ad.hip holds the form on dual numbers)
__device__ __forceinline__ void snap_put(GPTR(double) sn, unsigned c, double val) { stg(sn, c, val); }

// ---- face normals of a cell from its eight corner nodes, the formulas (and operand order) of metric_block
// (adjointExtra.F90:176-268, k_metric in kernels_geom.hip): a marching thread loads the two nodes (i, j, k) and (i, j-1, k) of
// its column per plane, takes the column i-1 by DPP and keeps the plane below: 6 loads instead of the 12 of sI, sJ(j-1), sJ, sK
struct NgNodes { double a[3], b[3]; };          // x(i, j, k) and x(i, j-1, k)

__device__ __forceinline__ void ngx_load_x(GPTR(const double) x, unsigned c, unsigned nb8, unsigned sj, NgNodes& n)
{
#pragma unroll
    for (int d = 0; d < 3; ++d) { n.a[d] = ldg(x, c + d * nb8); n.b[d] = ldg(x, c - sj + d * nb8); }
}

__device__ __forceinline__ void ngx_cross(double fact, const double p1[3], const double p2[3], const double q1[3], const double q2[3],
                                          double s[3])
{
    const double v1x = p1[0] - p2[0], v1y = p1[1] - p2[1], v1z = p1[2] - p2[2];
    const double v2x = q1[0] - q2[0], v2y = q1[1] - q2[1], v2z = q1[2] - q2[2];
    s[0] = fact * (v1y * v2z - v1z * v2y);
    s[1] = fact * (v1z * v2x - v1x * v2z);
    s[2] = fact * (v1x * v2y - v1y * v2x);
}

// sK of the node plane N alone (the plane below the first cell plane of a march)
__device__ __forceinline__ void ngx_normal_k(double fact, const NgNodes& N, double nK[3])
{
    double Na1[3], Nb1[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { Na1[d] = lane_up1(N.a[d]); Nb1[d] = lane_up1(N.b[d]); }
    ngx_cross(fact, N.a, Nb1, Na1, N.b, nK);                       // v1 = x(i,j,k) - x(l,m,k) ; v2 = x(l,j,k) - x(i,m,k)
}

// normals stored at the cell (i, j, k): sI, sJ(j-1), sJ, sK from the node planes k-1 (P) and k (N)
__device__ __forceinline__ void ngx_normals(double fact, const NgNodes& P, const NgNodes& N, double nI[3], double nJm[3], double nJ[3],
                                            double nK[3])
{
    double Na1[3], Nb1[3], Pa1[3], Pb1[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { Na1[d] = lane_up1(N.a[d]); Nb1[d] = lane_up1(N.b[d]); Pa1[d] = lane_up1(P.a[d]); Pb1[d] = lane_up1(P.b[d]); }
    ngx_cross(fact, P.a, N.b, N.a, P.b, nI);                       // v1 = x(i,j,n) - x(i,m,k) ; v2 = x(i,j,k) - x(i,m,n)
    ngx_cross(fact, P.a, Na1, Pa1, N.a, nJ);                       // v1 = x(i,j,n) - x(l,j,k) ; v2 = x(l,j,n) - x(i,j,k)
    ngx_cross(fact, P.b, Nb1, Pb1, N.b, nJm);                      // the same one row below
    ngx_cross(fact, N.a, Nb1, Na1, N.b, nK);
}


// coloured finite-difference Jacobian (kernels_jac.hip)
struct JacSpec {
    int nStencil, nState, lStart;     // lStart: first state variable (0-based) of the matrix
    int ca, cb, cc, cn, cm;           // colour(i,j,k) = (ca (i mod cm) + cb (j mod cm) + cc (k mod cm)) mod cn   (0-based)
    int st[33][3];                    // stencil offsets (row = perturbed cell + offset)
    int sc[33];                       // (ca di + cb dj + cc dk) mod cn of every entry (linear colourings)
};
void launch_fd_state(const BlkView& b, const double* wref, int l, int col, const JacSpec& J, double delta, hipStream_t s);
void launch_fd_state_closures(const BlkView& b, const double* wref, int l, int col, const JacSpec& J, double delta, const KParams& kp,
                              hipStream_t s, bool onlyL = false);
void launch_fd_snap(const BlkView& b, double* snap, const JacSpec& J, double turbResScale, hipStream_t s);
void launch_fd_scatter(const BlkView& b, const double* snap, double* jac, int l, const JacSpec& J, const double* dwref, double deltaInv,
                       hipStream_t s);
void launch_jac_rows(const BlkView& b, const double* jac, double* out, int nState, int nStencil, int k0, int nk, hipStream_t s);
void launch_fd_copy(const BlkView& b, double* dst, const double* src, int ncomp, hipStream_t s);
void launch_closures_halo(const BlkView& b, const KParams& kp, hipStream_t s);
void launch_fd_extract(const BlkView& b, double* dwref, double* jac, int l, int col, const JacSpec& J, double deltaInv, double turbResScale,
                       hipStream_t s);

// The level-batched launches fold (block slot, plane) into gridDim.z, which HIP limits to 65535: a launcher whose level has more
// slots than fit calls itself on consecutive slot ranges (the kernels index the table relative to the pointer they get).
extern int g_max_grid_z;          // 65535; tuning "max_grid_z" lowers it for the tests
extern int g_ra_pcr;              // kernels_smooth.hip, tuning "ra_pcr"
extern int g_dadi_pcr;            // kernels_smooth.hip, tuning "dadi_pcr"
inline int level_slots_per_launch(int planes)
{
    const int per = g_max_grid_z / (planes > 0 ? planes : 1);
    return per > 0 ? per : 1;
}
#define LEVEL_SPLIT(nslots, planes, CALL)                                                         \
    {                                                                                             \
        const int per_ = level_slots_per_launch(planes);                                          \
        if ((nslots) > per_) {                                                                    \
            for (int s0_ = 0; s0_ < (nslots); s0_ += per_) {                                      \
                const int n_ = ((nslots) - s0_ < per_) ? (nslots) - s0_ : per_;                   \
                CALL;                                                                             \
            }                                                                                     \
            return;                                                                               \
        }                                                                                         \
    }

// ---- kernel launchers (one translation unit per kernel family) -------------
void launch_time_step_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const KParams& kp, hipStream_t s);
void launch_entropy(const BlkView& b, hipStream_t s);
void launch_inviscid_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const KParams& kp, hipStream_t s);
void launch_initres(const BlkView& b, const KParams& kp, int l0, int l1, hipStream_t s);
void launch_viscous(const BlkView& b, const KParams& kp, hipStream_t s);
int viscous_is_tiled();
void launch_sa_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s, bool solve);
void launch_visc_march_approx(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s);
bool euler_march_radii_capable(const KParams& kp);
void launch_visc_gf(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, bool storeGrad, hipStream_t s);
void adf_note_snap(int bits);   // api.hip: a launcher reports that its kernel wrote the flow (1) / turbulence (2) snapshot entries (KParams::snapTab)
void adf_note_rvec(int bits);   // api.hip: a launcher reports that its kernel wrote the flow (1) / turbulence (2) part of kp.rvec
int adf_round_size();          // api.hip: workgroups of a marching kernel resident at a time (2 x CUs)
void adf_phase_mark(int i);    // api.hip: optional HIP event between the phases of blocketteRes
void launch_coarse_coordinates_level(const BlkView* ctab, const BlkView* ftab, int nslots, int nx, int ny, int nz, hipStream_t s);
void launch_xhalo_level(const BlkView* tab, int nslots, int nx, int ny, int nz, hipStream_t s);
void launch_xhalo_symm(const BlkView* tab, const BcEntry* ent, const int* order, const BcPhase& ph, hipStream_t s);
void launch_wall_stress(const BlkView* tab, const BcEntry* ent, const int* order, const BcPhase& ph, const KParams& kp, bool formGrad,
                        hipStream_t s);
void launch_viscous_approx(const BlkView& b, const KParams& kp, hipStream_t s);
void launch_face_vectors(const BlkView& b, hipStream_t s);
void launch_sa_residual_level(const BlkView* tab, int nslots, int nx, int ny, int nz, const KParams& kp, hipStream_t s);
void launch_sa_solve_level(const BlkView* tab, int nslots, int nx, int ny, int nz, const KParams& kp, hipStream_t s, bool marchRes = false);
void launch_rk_save_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, hipStream_t s);
void launch_etot_owned(const BlkView& b, double gammaConstant, hipStream_t s);
void launch_scale_dw_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, double factor, int timesVol, hipStream_t s);
void launch_source_terms(const BlkView* tab, const int* blk, const long* off, int n, const double Ffact[3], double Qfact, int withBlank,
                         hipStream_t s);
void launch_low_speed_precond_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const KParams& kp, hipStream_t s);
void launch_stage_update_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const KParams& kp, double scale,
                               int fromWn, hipStream_t s);
void launch_res_averaging_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const KParams& kp, hipStream_t s,
                                double scaleDtl = 0.0);
void launch_dadi_level(const BlkView* tab, int nslots, int nx, int ny, int nz, const KParams& kp, hipStream_t s, bool withUpdate = false);
void launch_halo_copy(const BlkView* tab, const int* donorBlk, const long* donorOff, const int* haloBlk, const long* haloOff,
                      int n, unsigned mask, hipStream_t s);
void launch_periodic(const BlkView* tab, const int* blk, const long* off, int n, const double rotMatrix[9], const double rotCenter[3],
                     const double translation[3], int coor, hipStream_t s);
void launch_halo_pack(const BlkView* tab, const int* blk, const long* off, int n, unsigned mask, double* buf, hipStream_t s);
void launch_halo_unpack(const BlkView* tab, const int* blk, const long* off, int n, unsigned mask, const double* buf,
                        hipStream_t s);
void launch_inviscid_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s);
int inviscid_march_enabled();
bool launch_roe_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s);
// kernels_pc_march.hip: first-order Roe + thin-layer viscous flux in one march (the mean-flow residual of the preconditioner matrix)
void launch_pc_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, int kch, hipStream_t s);
bool roe_march_takes(const KParams& kp);
void launch_euler_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s);
void euler_march_tiles(const BlkView& b, int* ntx, int* nty, int* ntz);
void launch_restrict_level(const BlkView* ctab, const BlkView* ftab, int nslots, int nx, int ny, int nz, const KParams& kp, hipStream_t s);
void launch_store_entry_state_level(const BlkView* ctab, int nslots, int nx, int ny, int nz, hipStream_t s);
void launch_forcing_level(const BlkView* ctab, int nslots, int nx, int ny, int nz, double fcoll, hipStream_t s);
void launch_corrections_level(const BlkView* ctab, int nslots, int nx, int ny, int nz, hipStream_t s);
void launch_prolong_update_level(const BlkView* ftab, const BlkView* ctab, int nslots, int nx, int ny, int nz, const KParams& kp,
                                 hipStream_t s);
void launch_set_w(const BlkView& b, const double* vec, double turbFloor, hipStream_t s);
void launch_get_r(const BlkView& b, double* vec, double turbScale, double* sums, hipStream_t s);
void launch_closures(const BlkView& b, const KParams& kp, hipStream_t s);
// level-batched forms (blockIdx.z = slot * planes + plane): one launch for every block of a level
void launch_closures_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const KParams& kp, hipStream_t s,
                           int* floored = nullptr);
void launch_set_w_closures_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const double* vec, double turbFloor,
                                 const KParams& kp, int* floored, hipStream_t s);
void launch_set_w_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const double* vec, double turbFloor, hipStream_t s);
void launch_get_r_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, double* vec, double turbScale, double* sums, hipStream_t s);
void launch_entropy_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, hipStream_t s);
void launch_etot_owned_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, double gammaConstant, hipStream_t s,
                             const int* onlyIf = nullptr);
void launch_initres_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const KParams& kp, int l0, int l1, hipStream_t s);
void launch_res_norms(const BlkView& b, int nvar, double* out, hipStream_t s);
#include <vector>
void launch_apply_all_bc(const BlkView* tab, const BcEntry* ent, const int* order, const std::vector<BcPhase>& flow,
                         const KParams& kp, int second, int eulerWallTreatment, int viscWallTreatment, int outflowTreatment, int hScalingInlet,
                         hipStream_t s);
void launch_turb_bc_treatment(const BlkView* tab, int nslots, long maxFace, const BcEntry* ent, const int* order,
                              const std::vector<BcPhase>& ordinal, const KParams& kp, hipStream_t s);
void launch_apply_turb_bc(const BlkView* tab, const BcEntry* ent, const int* order, const std::vector<BcPhase>& ordinal,
                          const KParams& kp, int second, hipStream_t s);
void launch_bc_coarse_corrections(const BlkView* tab, const BcEntry* ent, const int* order, const std::vector<BcPhase>& ordinal,
                                  double fact, hipStream_t s);
void launch_corner_row_halos_level(const BlkView* tab, int nslots, const KParams& kp, hipStream_t s);
void launch_volume_metric(const BlkView& b, int rightHanded, hipStream_t s);
void launch_boundary_normals(const BlkView& b, const BcFaceDev* faces, int nBocos, hipStream_t s);
void launch_wall_distance(const BlkView& b, const int* ind, const double* uv, const double* xSurf, hipStream_t s);

// ---- forward-mode twins of the gather kernels (kernels_ad.hip)
// entry points of kernels_ad.hip (BlkView-layout views whose array pointers lead to dual numbers, blkview_def.h)
void ad_launch_from_real(const double* src, void* dst, long n, hipStream_t s);
void ad_launch_value(const void* src, double* dst, long n, int deriv, hipStream_t s);
void ad_launch_snap(const BlkView& b, const void* dwd, double* snap, const JacSpec& J, double turbResScale, hipStream_t s);
void ad_launch_closures_halo(const BlkView& adv, const KParams& kp, hipStream_t s);
void ad_launch_apply_all_bc(const BlkView* tab, const BcEntry* ent, const int* order, const std::vector<BcPhase>& flow, const KParams& kp,
                            int second, int eulerWallTreatment, int viscWallTreatment, int outflowTreatment, int hScalingInlet, hipStream_t s);
void ad_launch_turb_bc(const BlkView* tab, int nslots, long maxFace, const BcEntry* ent, const int* order, const std::vector<BcPhase>& ordinal,
                       const KParams& kp, int second, hipStream_t s);
void ad_launch_time_step_level(const BlkView* tab, int n, int nx, int ny, int nz, const KParams& kp, hipStream_t s);
void ad_launch_entropy_level(const BlkView* tab, int n, int nx, int ny, int nz, hipStream_t s);
void ad_launch_sa_residual_level(const BlkView* tab, int n, int nx, int ny, int nz, const KParams& kp, hipStream_t s);
void ad_launch_inviscid_level(const BlkView* tab, int n, int nx, int ny, int nz, const KParams& kp, hipStream_t s);
void ad_launch_viscous(const BlkView& adv, const KParams& kp, hipStream_t s);
void ad_launch_viscous_approx(const BlkView& adv, const KParams& kp, hipStream_t s);
void ad_launch_seed_closures(const BlkView& real, const BlkView& adv, int l, int col, const JacSpec& J, const KParams& kp, hipStream_t s,
                             bool onlyL);
void ad_launch_pc_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, int kch, hipStream_t s);
void ad_launch_sa_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s);
void ad_launch_visc_gf(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s);
bool ad_launch_roe_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s);
void ad_launch_inviscid_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s);
void ad_launch_selftest_math(int which, const double* x, const double* a, long n, double* y, double* dy, hipStream_t s);
void ad_launch_visc_march_approx(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s);
