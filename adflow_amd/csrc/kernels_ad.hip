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
__device__ __forceinline__ Dual rcp_nr(const Dual& b) { return 1.0 / b; }
__device__ __forceinline__ Dual rsq_nr(const Dual& x) { return 1.0 / sqrt(x); }
__device__ __forceinline__ Dual fastdiv(const Dual& a, const Dual& b) { return a / b; }
__device__ __forceinline__ Dual fastdiv(double a, const Dual& b) { return a / b; }
__device__ __forceinline__ Dual fastdiv(const Dual& a, double b) { return a / b; }
__device__ __forceinline__ Dual fastsqrt(const Dual& x) { return sqrt(x); }
__device__ __forceinline__ Dual fast_root6(const Dual& x) { return pow(x, 1.0 / 6.0); }
__device__ __forceinline__ Dual fast_exp_neg(const Dual& x) { return exp(x); }
__device__ __forceinline__ Dual fast_powa(const Dual& x, double a) { return pow(x, a); }

// ---- conversions between the library's arrays and the dual arrays (written before `double` changes its meaning)
__global__ void k_ad_from_real(const double* __restrict__ src, Dual* __restrict__ dst, long n)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = Dual(src[t], 0.0);
}
// w <- (w, 1 where the cell has colour `col` and the component is l, else 0): the seed of one pass (adjointUtils.F90:330-347)
__global__ void k_ad_seed(BlkView b, Dual* __restrict__ wd, int l, int col, JacSpec J)
{
    const int i = blockIdx.x * 64 + threadIdx.x - 14;
    const int j = blockIdx.y * 4 + threadIdx.y;
    const int k = blockIdx.z;
    if (i < 0 || i > b.ib || j > b.jb) return;
    const long c = b.idx(i, j, k);
    const bool hit = ((J.ca * (i % J.cm) + J.cb * (j % J.cm) + J.cc * (k % J.cm)) % J.cn) == col;
    for (int m = 0; m < b.nw; ++m) wd[c + m * b.nbox] = Dual(b.w[c + m * b.nbox], (hit && m == l) ? 1.0 : 0.0);
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
__global__ __launch_bounds__(256) void k_closures_halo(BlkView b, KParams kp)
{
    const int i = blockIdx.x * 64 + threadIdx.x - 14;
    const int j = blockIdx.y * 4 + threadIdx.y;
    const int k = blockIdx.z;
    if (i < 0 || i > b.ib || j > b.jb) return;
    const long c = b.idx(i, j, k), nb = b.nbox;
    const double rho = b.w[c], u = b.w[c + nb], v = b.w[c + 2 * nb], w = b.w[c + 3 * nb];
    const adf_real8 gm1 = kp.gammaConstant - 1.0;
    double p = gm1 * (b.w[c + 4 * nb] - 0.5 * rho * (u * u + v * v + w * w));
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
        const double rnuSA = b.w[c + 5 * nb] * rho;
        const double chi = rnuSA / rlv;
        const double chi3 = chi * chi * chi;
        b.rev[c] = chi3 / (chi3 + cv13) * rnuSA;
    }
}

#include "kernels_inviscid.hip"
#include "kernels_timestep.hip"
#include "kernels_bc.hip"
#include "kernels_sa.hip"
#include "kernels_viscous.hip"

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
void ad_launch_seed(const BlkView& b, void* wd, int l, int col, const JacSpec& J, hipStream_t s)
{
    hipLaunchKernelGGL(k_ad_seed, dim3((b.ib + 15 + 63) / 64, (b.jb + 4) / 4, b.kb + 1), dim3(64, 4, 1), 0, s, b, (Dual*)wd, l, col, J);
}
void ad_launch_snap(const BlkView& b, const void* dwd, double* snap, const JacSpec& J, double turbResScale, hipStream_t s)
{
    hipLaunchKernelGGL(k_ad_snap, dim3((b.nx + 63) / 64, (b.ny + 3) / 4, b.nz), dim3(64, 4, 1), 0, s, b, (const Dual*)dwd, snap, J, turbResScale);
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
#undef ADV
