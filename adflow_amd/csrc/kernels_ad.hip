// Forward-mode (dual-number) twins of the gather kernels of the residual: SURVEY.md 8(f) #4, the `useAD = T` branch of
// adjointUtils::setupStateResidualMatrix (adjointUtils.F90:227-409) = masterRoutines::block_res_state_d (:1285-1393):
//   computePressureSimple_d / computeLamViscosity_d / computeEddyViscosity_d (includeHalos), bcTurbTreatment_d,
//   applyAllTurbBCThisBlock_d, applyAllBC_block_d, timeStep_block_d, the SA source / advection / diffusion _d routines,
//   inviscidCentralFlux_d, the (approximate) dissipation _d routines, computeSpeedOfSoundSquared_d, allNodalGradients_d,
//   viscousFlux_d | viscousFluxApprox_d, sumDwAndFw_d, resScale_d.
// The reference gets these from Tapenade (src/adjoint/outputForward/*.f90).  Here the SAME kernel sources the finite-difference
// assembly runs (kernels_inviscid / timestep / bc / sa / viscous: the cell-gather forms, one thread per cell, no lane exchange) are
// compiled a second time inside namespace adj with `double` standing for the dual number of dual.h: state, residual and every
// intermediate carry value + derivative with respect to the one seed direction of the pass; options (KParams) and the boundary data
// stay plain doubles.  BlkViewAD is BlkView with dual arrays (blkview_def.h: same layout), filled by the host through BlkView.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "internal.h"
#include "dual.h"

// ---- the helpers of internal.h for dual arguments (the fast reciprocal / root forms are plain operations here)
// (round 5: value by the same fast form as the plain build, derivative from it -- the IEEE divisions and roots of `1.0 / b` cost the
//  dual kernels 3.5 x the transcendental seeds of the plain ones.  The derivative is formed factor by factor, never through r^2 or
//  r^3 alone: the entropy fix of roe_face.h takes the reciprocal of (z1l + z1r) max(eta, 1e-290), whose square overflows where the
//  two states of a face are equal -- found by tests/fuzz_parity.py --jac on an extrapolated halo)
__device__ __forceinline__ Dual rcp_nr(const Dual& b)
{
    const double r = rcp_nr(b.v);
    return Dual(r, -(r * b.d) * r);
}
__device__ __forceinline__ Dual rsq_nr(const Dual& x)
{
    const double r = rsq_nr(x.v);
    return Dual(r, (((-0.5 * r) * x.d) * r) * r);
}
__device__ __forceinline__ Dual fastdiv(const Dual& a, const Dual& b)
{
    const double r = rcp_nr(b.v), q = a.v * r;
    return Dual(q, (a.d - q * b.d) * r);
}
__device__ __forceinline__ Dual fastdiv(double a, const Dual& b)
{
    const double r = rcp_nr(b.v), q = a * r;
    return Dual(q, -(q * b.d) * r);
}
__device__ __forceinline__ Dual fastdiv(const Dual& a, double b)
{
    const double r = rcp_nr(b);
    return Dual(a.v * r, a.d * r);
}
__device__ __forceinline__ Dual fastsqrt(const Dual& x)        // sqrt(0): derivative 0 (dual.h)
{
    const double r = rsq_nr(fmax(x.v, 1.e-300));
    return Dual(x.v * r, x.v == 0.0 ? 0.0 : (0.5 * r) * x.d);
}
__device__ __forceinline__ Dual fast_root6(const Dual& x) { return pow(x, 1.0 / 6.0); }
__device__ __forceinline__ Dual fast_exp_neg(const Dual& x) { return exp(x); }
__device__ __forceinline__ Dual fast_powa(const Dual& x, double a) { return pow(x, a); }
// memory and lane moves of the marching form (kernels_pc_march.hip)
// The marching kernels address every array of a block with ONE byte offset of 8-byte elements (internal.h: ldg / stg); an array of
// dual numbers has 16-byte elements at the same indices, so its forms of ldg / stg DOUBLE the offset (valid while a component of a
// block stays below 2 GiB of doubles) -- the same kernel source then addresses plain geometry and dual state side by side.
typedef double adf_v2 __attribute__((vector_size(16)));     // (value, derivative) as ONE 16-byte access
__device__ __forceinline__ Dual ldg(GPTR(const Dual) base, unsigned byteoff8)
{
    const adf_v2 t = *(GPTR(const adf_v2))((GPTR(const char))base + 2u * byteoff8);
    return Dual(t[0], t[1]);
}
__device__ __forceinline__ void stg(GPTR(Dual) base, unsigned byteoff8, const Dual& v)
{
    const adf_v2 t = {v.v, v.d};
    *(GPTR(adf_v2))((GPTR(char))base + 2u * byteoff8) = t;
}
// the snapshot entry of a forward-mode pass: the derivative part (k_ad_snap; no reference residual, no step)
__device__ __forceinline__ void snap_put(GPTR(double) sn, unsigned c, const Dual& val) { stg(sn, c, val.d); }
__device__ __forceinline__ int adf_hiword(const Dual& a) { return adf_hiword(a.v); }
__device__ __forceinline__ Dual adf_fma(const Dual& a, const Dual& b, const Dual& c) { return a * b + c; }
__device__ __forceinline__ Dual adf_fma(const Dual& a, const Dual& b, double c) { return a * b + c; }
__device__ __forceinline__ Dual lane_up1(const Dual& a) { return Dual(lane_up1(a.v), lane_up1(a.d)); }
__device__ __forceinline__ Dual lane_dn1(const Dual& a) { return Dual(lane_dn1(a.v), lane_dn1(a.d)); }

// ---- conversions between the library's arrays and the dual arrays (written before `double` changes its meaning)
__global__ void k_ad_from_real(const double* __restrict__ src, Dual* __restrict__ dst, long n)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = Dual(src[t], 0.0);
}
// derivative part of the scaled residual -> the dense snapshot of the pass (resScale_d + the copy into dw_deriv, adjointUtils.F90:384-388)
__global__ void k_ad_snap(BlkView b, const Dual* __restrict__ dwd, double* __restrict__ snap, JacSpec J, double turbResScale)
{
    const int i = blockIdx.x * 64 + threadIdx.x + 2;
    const int j = blockIdx.y * 4 + threadIdx.y + 2;
    const int k = blockIdx.z + 2;
    if (i > b.il || j > b.jl) return;
    const long c = b.idx(i, j, k);
    const double ovol = 1.0 / b.volRef[c];
    for (int m = 0; m < J.nState; ++m) {
        const int ll = J.lStart + m;
        snap[c + m * b.nbox] = dwd[c + ll * b.nbox].d * ovol * (ll >= 5 ? turbResScale : 1.0);
    }
}
// value part of a dual array back into a plain one (tests)
__global__ void k_ad_value(const Dual* __restrict__ src, double* __restrict__ dst, long n, int deriv)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = deriv ? src[t].d : src[t].v;
}

// adflow_gpu_selftest_math: the fast reciprocal / root / power forms of internal.h and their dual-number forms above, one argument per
// thread (the kernel-logic emulator replaces them with libm: only a device run sees them)
__global__ void k_selftest_math(int which, const double* __restrict__ x, const double* __restrict__ a, long n, double* __restrict__ y,
                                double* __restrict__ dy)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double xv = x[t], av = a[t];
    const Dual xd(xv, 1.0);
    double v = 0.0;
    Dual d(0.0, 0.0);
    switch (which) {
    case 0: v = rcp_nr(xv); d = rcp_nr(xd); break;
    case 1: v = rsq_nr(xv); d = rsq_nr(xd); break;
    case 2: v = fastsqrt(xv); d = fastsqrt(xd); break;
    case 3: v = fast_root6(xv); d = fast_root6(xd); break;
    case 4: v = fast_exp_neg(xv); d = fast_exp_neg(xd); break;
    case 5: v = fast_powa(xv, av); d = fast_powa(xd, av); break;
    case 6: v = fastdiv(av, xv); d = fastdiv(av, xd); break;
    case 7: v = fastdiv(xv, av); d = fastdiv(xd, Dual(av, 0.0)); break;
    default: break;
    }
    y[t] = v;
    dy[2 * t] = d.v;
    dy[2 * t + 1] = d.d;
}
void ad_launch_selftest_math(int which, const double* x, const double* a, long n, double* y, double* dy, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_selftest_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, which, x, a, n, y, dy);
}

extern int g_march_kch, g_roe_march;
namespace adj {
#define ADF_AD_BUILD 1
#define double Dual
#define ADF_BLKVIEW BlkViewAD
#define ADF_GEOM adf_real8
#include "blkview_def.h"
#undef ADF_GEOM
#undef ADF_BLKVIEW
#define BlkView BlkViewAD

// closures of block_res_state_d: pressure on 0..ib, laminar / eddy viscosity on 1..ie (includeHalos = .True.)
__device__ __forceinline__ void closures_halo_cell(const BlkView& b, const KParams& kp, int i, int j, int k, long c, const double wv[6])
{
    const double rho = wv[0], u = wv[1], v = wv[2], w = wv[3];
    const adf_real8 gm1 = kp.gammaConstant - 1.0;
    double p = gm1 * (wv[4] - 0.5 * rho * (u * u + v * v + w * w));
    p = fmax(p, 1.e-4 * kp.pInfCorr);
    b.p[c] = p;
    if (!kp.viscous || i < 1 || i > b.ie || j < 1 || j > b.je || k < 1 || k > b.ke) return;
    const adf_real8 muSuth = kp.muSuthDim / kp.muRef, TSuth = kp.TSuthDim / kp.TRef, SSuth = kp.SSuthDim / kp.TRef;
    const double T = p / (kp.RGas * rho);
    const double tt = T / TSuth;
    const double rlv = muSuth * ((TSuth + SSuth) / (T + SSuth)) * (tt * sqrt(tt));
    b.rlv[c] = rlv;
    if (kp.eddyModel) {
        const adf_real8 cv13 = kp.sa_cv1 * kp.sa_cv1 * kp.sa_cv1;
        const double rnuSA = wv[5] * rho;
        const double chi = rnuSA / rlv;
        const double chi3 = chi * chi * chi;
        b.rev[c] = chi3 / (chi3 + cv13) * rnuSA;
    }
}
__global__ __launch_bounds__(256) void k_closures_halo(BlkView b, KParams kp)
{
    const int i = blockIdx.x * 64 + threadIdx.x - 14;
    const int j = blockIdx.y * 4 + threadIdx.y;
    const int k = blockIdx.z;
    if (i < 0 || i > b.ib || j > b.jb) return;
    const long c = b.idx(i, j, k), nb = b.nbox;
    double wv[6];
    for (int m = 0; m < 6; ++m) wv[m] = (m < b.nw) ? b.w[c + m * nb] : 0.0;
    closures_halo_cell(b, kp, i, j, k, c, wv);
}
// the seed of one pass (adjointUtils.F90:330-347) and the closures from it in ONE pass over the state: w <- (w of the library, 1 where
// the cell has colour `col` and the component is l), then pressure and viscosities from the values in registers (round 5: two launches read
// the 96 B per cell of the dual state back that the first had just written)
// onlyL: inside the sweep of one state variable over the colours only component l of the dual state changes (the others keep their
// values and a zero seed): 16 instead of 96 B per cell written after the first colour
__global__ __launch_bounds__(256) void k_seed_closures(BlkView b, const adf_real8* __restrict__ wsrc, int l, int col, JacSpec J, KParams kp,
                                                       int onlyL)
{
    const int i = blockIdx.x * 64 + threadIdx.x - 14;
    const int j = blockIdx.y * 4 + threadIdx.y;
    const int k = blockIdx.z;
    if (i < 0 || i > b.ib || j > b.jb) return;
    const long c = b.idx(i, j, k), nb = b.nbox;
    const bool hit = ((J.ca * (i % J.cm) + J.cb * (j % J.cm) + J.cc * (k % J.cm)) % J.cn) == col;
    double wv[6];
    for (int m = 0; m < 6; ++m) {
        if (m < b.nw) {
            wv[m] = Dual(wsrc[c + m * nb], (hit && m == l) ? 1.0 : 0.0);
            if (!onlyL || m == l) b.w[c + m * nb] = wv[m];
        } else
            wv[m] = 0.0;
    }
    closures_halo_cell(b, kp, i, j, k, c, wv);
}

#include "kernels_inviscid.hip"
#include "kernels_timestep.hip"
#include "kernels_bc.hip"
#include "kernels_sa.hip"
#include "kernels_viscous.hip"
#include "kernels_pc_march.hip"
#include "kernels_sa_march.hip"
#include "kernels_roe_march.hip"
#include "kernels_inviscid_march.hip"

#undef BlkView
#undef double
}  // namespace adj

// ---- host-callable entry points.  api.hip knows BlkView only; a table of BlkView-layout entries whose array pointers lead to dual
// arrays IS a table of BlkViewAD (blkview_def.h), so the casts below change the static type, not the data.
#define ADV(p) reinterpret_cast<const adj::BlkViewAD*>(p)
void ad_launch_from_real(const double* src, void* dst, long n, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_ad_from_real, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, (Dual*)dst, n);
}
void ad_launch_value(const void* src, double* dst, long n, int deriv, hipStream_t s)
{
    if (n > 0) hipLaunchKernelGGL(k_ad_value, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const Dual*)src, dst, n, deriv);
}
void ad_launch_snap(const BlkView& b, const void* dwd, double* snap, const JacSpec& J, double turbResScale, hipStream_t s)
{
    hipLaunchKernelGGL(k_ad_snap, dim3((b.nx + 63) / 64, (b.ny + 3) / 4, b.nz), dim3(64, 4, 1), 0, s, b, (const Dual*)dwd, snap, J, turbResScale);
}
void ad_launch_seed_closures(const BlkView& real, const BlkView& adv, int l, int col, const JacSpec& J, const KParams& kp, hipStream_t s,
                             bool onlyL)
{
    hipLaunchKernelGGL(adj::k_seed_closures, dim3((adv.ib + 15 + 63) / 64, (adv.jb + 4) / 4, adv.kb + 1), dim3(64, 4, 1), 0, s, *ADV(&adv),
                       real.w, l, col, J, kp, onlyL ? 1 : 0);
}
void ad_launch_closures_halo(const BlkView& adv, const KParams& kp, hipStream_t s)
{
    hipLaunchKernelGGL(adj::k_closures_halo, dim3((adv.ib + 15 + 63) / 64, (adv.jb + 4) / 4, adv.kb + 1), dim3(64, 4, 1), 0, s, *ADV(&adv), kp);
}
void ad_launch_apply_all_bc(const BlkView* tab, const BcEntry* ent, const int* order, const std::vector<BcPhase>& flow, const KParams& kp,
                            int second, int eulerWallTreatment, int viscWallTreatment, int outflowTreatment, int hScalingInlet, hipStream_t s)
{
    adj::launch_apply_all_bc(ADV(tab), ent, order, flow, kp, second, eulerWallTreatment, viscWallTreatment, outflowTreatment, hScalingInlet, s);
}
void ad_launch_turb_bc(const BlkView* tab, int nslots, long maxFace, const BcEntry* ent, const int* order, const std::vector<BcPhase>& ordinal,
                       const KParams& kp, int second, hipStream_t s)
{
    adj::launch_turb_bc_treatment(ADV(tab), nslots, maxFace, ent, order, ordinal, kp, s);
    adj::launch_apply_turb_bc(ADV(tab), ent, order, ordinal, kp, second, s);
}
void ad_launch_time_step_level(const BlkView* tab, int n, int nx, int ny, int nz, const KParams& kp, hipStream_t s)
{
    adj::launch_time_step_level(ADV(tab), n, nx, ny, nz, kp, s);
}
void ad_launch_entropy_level(const BlkView* tab, int n, int nx, int ny, int nz, hipStream_t s) { adj::launch_entropy_level(ADV(tab), n, nx, ny, nz, s); }
void ad_launch_sa_residual_level(const BlkView* tab, int n, int nx, int ny, int nz, const KParams& kp, hipStream_t s)
{
    adj::launch_sa_residual_level(ADV(tab), n, nx, ny, nz, kp, s);
}
void ad_launch_inviscid_level(const BlkView* tab, int n, int nx, int ny, int nz, const KParams& kp, hipStream_t s)
{
    adj::launch_inviscid_level(ADV(tab), n, nx, ny, nz, kp, s);
}
void ad_launch_viscous(const BlkView& adv, const KParams& kp, hipStream_t s) { adj::launch_viscous(*ADV(&adv), kp, s); }
void ad_launch_viscous_approx(const BlkView& adv, const KParams& kp, hipStream_t s) { adj::launch_viscous_approx(*ADV(&adv), kp, s); }
void ad_launch_pc_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, int kch, hipStream_t s)
{
    adj::launch_pc_march(ADV(tab), tiles, ntiles, kp, kch, s);
}
void ad_launch_sa_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s)
{
    adj::launch_sa_march(ADV(tab), tiles, ntiles, kp, s, false);
}
void ad_launch_visc_gf(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s)
{
    adj::launch_visc_gf(ADV(tab), tiles, ntiles, kp, false, s);
}
bool ad_launch_roe_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s)
{
    return adj::launch_roe_march(ADV(tab), tiles, ntiles, kp, s);
}
void ad_launch_visc_march_approx(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s)
{
    adj::launch_visc_march_approx(ADV(tab), tiles, ntiles, kp, s);
}
void ad_launch_inviscid_march(const BlkView* tab, const int4* tiles, int ntiles, const KParams& kp, hipStream_t s)
{
    adj::launch_inviscid_march(ADV(tab), tiles, ntiles, kp, s);
}
#undef ADV
