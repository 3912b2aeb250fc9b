// Newton-Krylov glue: PETSc state/residual vectors <-> block arrays, and the
// pointwise state closures that open blockette::blocketteRes.
//
// Reference semantics:
//   NKSolvers::setW      src/NKSolver/NKSolvers.F90:1331-1376  (AoS vector -> w, turbulence clipped at 1e-6*wInf)
//   NKSolvers::setRVec   src/NKSolver/NKSolvers.F90:1262-1329  (dw/volRef, turbulence * turbResScale, AoS)
//   nksolver::getRes     src/NKSolver/NKSolvers.F90:1413-1450  (same without turbResScale)
//   computePressureSimple / computeLamViscosity / computeEddyViscosity on the owned cells
//                        src/NKSolver/blockette.F90:199-203, flowUtils.F90:867-925,1201-1300, turbUtils.F90:657-720
// Vector order: block, k, j, i, variable fastest (NKSolvers.F90:1240-1253).
#include <algorithm>

#include "internal.h"

#define NK_BX 64
#define NK_BY 4

__device__ __forceinline__ void set_w_body(const BlkView& b, int kz, const double* __restrict__ vec, double turbFloor)
{
    const int i = blockIdx.x * NK_BX + threadIdx.x + 2;
    const int j = blockIdx.y * NK_BY + threadIdx.y + 2;
    const int k = kz + 2;
    if (i > b.il || j > b.jl || k > b.kl) return;
    const long c = b.idx(i, j, k);
    const long m = (((long)(k - 2) * b.ny + (j - 2)) * b.nx + (i - 2)) * b.nw;
    for (int l = 0; l < 5; ++l) b.w[c + l * b.nbox] = vec[m + l];
    if (b.nw > 5) b.w[c + 5 * b.nbox] = fmax(turbFloor, vec[m + 5]);
}

__global__ __launch_bounds__(NK_BX* NK_BY) void k_set_w(BlkView b, const double* __restrict__ vec, double turbFloor)
{
    set_w_body(b, (int)blockIdx.z, vec, turbFloor);
}

__global__ __launch_bounds__(NK_BX* NK_BY) void k_set_w_level(const BlkView* __restrict__ tab, int nzb, const double* __restrict__ vec,
                                                              double turbFloor)
{
    const BlkView& b = tab[blockIdx.z / nzb + 1];
    set_w_body(b, (int)(blockIdx.z % nzb), vec + b.vecOff, turbFloor);
}

// scale != 0: turbulence residual multiplied by `turbScale`; sums[0] += flow^2, sums[1] += turb^2
__device__ __forceinline__ void get_r_body(const BlkView& b, int kz, double* __restrict__ vec, double turbScale, double* __restrict__ sums)
{
    __shared__ double red[2][NK_BX * NK_BY];
    const int tid = threadIdx.x + NK_BX * threadIdx.y;
    const int i = blockIdx.x * NK_BX + threadIdx.x + 2;
    const int j = blockIdx.y * NK_BY + threadIdx.y + 2;
    const int k = kz + 2;
    double sf = 0.0, st = 0.0;
    if (i <= b.il && j <= b.jl && k <= b.kl) {
        const long c = b.idx(i, j, k);
        const long m = (((long)(k - 2) * b.ny + (j - 2)) * b.nx + (i - 2)) * b.nw;
        const double ovv = 1.0 / b.volRef[c];
        for (int l = 0; l < 5; ++l) {
            const double t = b.dw[c + l * b.nbox] * ovv;
            vec[m + l] = t;
            sf += t * t;
        }
        if (b.nw > 5) {
            const double t = b.dw[c + 5 * b.nbox] * ovv * turbScale;
            vec[m + 5] = t;
            st += t * t;
        }
    }
    if (!sums) return;
    red[0][tid] = sf;
    red[1][tid] = st;
    __syncthreads();
    for (int s = NK_BX * NK_BY / 2; s > 0; s >>= 1) {
        if (tid < s) {
            red[0][tid] += red[0][tid + s];
            red[1][tid] += red[1][tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        atomicAdd(&sums[0], red[0][0]);
        atomicAdd(&sums[1], red[1][0]);
    }
}

__global__ __launch_bounds__(NK_BX* NK_BY) void k_get_r(BlkView b, double* __restrict__ vec, double turbScale, double* __restrict__ sums)
{
    get_r_body(b, (int)blockIdx.z, vec, turbScale, sums);
}

__global__ __launch_bounds__(NK_BX* NK_BY) void k_get_r_level(const BlkView* __restrict__ tab, int nzb, double* __restrict__ vec,
                                                              double turbScale, double* __restrict__ sums)
{
    const BlkView& b = tab[blockIdx.z / nzb + 1];
    get_r_body(b, (int)(blockIdx.z % nzb), vec + b.vecOff, turbScale, sums);
}

// owned cells: p from (rho, v, rhoE) with the 1e-4*pInfCorr floor, Sutherland, SA eddy viscosity
// ETOT: also rhoE back from p, the pass whalo2 closes with on the owned cells (haloExchange.F90:178-196, k_etot_owned_level).  The
// exchange that follows in FormFunction_mf carries the energy of the VECTOR into the neighbours' halos and only then recomputes the
// owned one: where p was not floored the two agree (the same expression of the same values), so the energy is rewritten here and the
// pass over w, p is saved; a cell whose pressure hit the floor keeps the vector's energy for the exchange and raises *floored, on
// which the pass behind the exchange then runs as before
template <bool ETOT = false>
__device__ __forceinline__ void closures_cell(const BlkView& b, int i, int j, int k, const KParams& kp, int* __restrict__ floored);
template <bool ETOT = false>
__device__ __forceinline__ void closures_body(const BlkView& b, int kz, const KParams& kp, int* __restrict__ floored = nullptr)
{
    const int i = blockIdx.x * NK_BX + threadIdx.x + 2;
    const int j = blockIdx.y * NK_BY + threadIdx.y + 2;
    const int k = kz + 2;
    if (i > b.il || j > b.jl || k > b.kl) return;
    closures_cell<ETOT>(b, i, j, k, kp, floored);
}
template <bool ETOT>
__device__ __forceinline__ void closures_cell(const BlkView& b, int i, int j, int k, const KParams& kp, int* __restrict__ floored)
{
    const long c = b.idx(i, j, k);
    const long nb = b.nbox;
    const double rho = b.w[c], u = b.w[c + nb], v = b.w[c + 2 * nb], w = b.w[c + 3 * nb];
    const double gm1 = kp.gammaConstant - 1.0;
    const double v2 = u * u + v * v + w * w;
    double p = gm1 * (b.w[c + 4 * nb] - 0.5 * rho * v2);
    const double pFloor = 1.e-4 * kp.pInfCorr;
    const bool hitFloor = !(p >= pFloor);
    p = fmax(p, pFloor);
    b.p[c] = p;
    if (ETOT) {
        if (hitFloor) *floored = 1;
        else {
            const double ovgm1 = 1.0 / (kp.gammaConstant - 1.0);
            b.w[c + 4 * nb] = ovgm1 * p + 0.5 * rho * v2;
        }
    }
    if (kp.viscous) {
        const double muSuth = kp.muSuthDim / kp.muRef, TSuth = kp.TSuthDim / kp.TRef, SSuth = kp.SSuthDim / kp.TRef;
        const double T = p / (kp.RGas * rho);
        const double tt = T / TSuth;
        const double rlv = muSuth * ((TSuth + SSuth) / (T + SSuth)) * (tt * sqrt(tt));
        b.rlv[c] = rlv;
        if (kp.eddyModel && kp.updateEddy) {
            const double cv13 = kp.sa_cv1 * kp.sa_cv1 * kp.sa_cv1;
            const double rnuSA = b.w[c + 5 * nb] * rho;
            const double chi = rnuSA / rlv;
            const double chi3 = chi * chi * chi;
            b.rev[c] = chi3 / (chi3 + cv13) * rnuSA;
        }
    }
}

__global__ __launch_bounds__(NK_BX* NK_BY) void k_closures(BlkView b, KParams kp) { closures_body(b, (int)blockIdx.z, kp); }

// setW followed by the closures of blocketteRes in one pass (FormFunction_mf: NKSolvers.F90:437-461 -> blockette.F90:199-203):
// each thread reads back only what it wrote itself
__global__ __launch_bounds__(NK_BX* NK_BY) void k_set_w_closures_level(const BlkView* __restrict__ tab, int nzb, const double* __restrict__ vec,
                                                                       double turbFloor, KParams kp, int* __restrict__ floored)
{
    const BlkView& b = tab[blockIdx.z / nzb + 1];
    set_w_body(b, (int)(blockIdx.z % nzb), vec + b.vecOff, turbFloor);
    closures_body<true>(b, (int)(blockIdx.z % nzb), kp, floored);
}

// ETOT: the closures of blocketteRes when its whalo2 follows (blockette.F90:199-246): the energy whalo2 would recompute on the owned
// cells is written here, as in k_set_w_closures_level
template <bool ETOT>
__global__ __launch_bounds__(NK_BX* NK_BY) void k_closures_level(const BlkView* __restrict__ tab, int nzb, KParams kp, int* __restrict__ floored)
{
    closures_body<ETOT>(tab[blockIdx.z / nzb + 1], (int)(blockIdx.z % nzb), kp, floored);
}

static dim3 nk_grid(const BlkView& b) { return dim3((b.nx + NK_BX - 1) / NK_BX, (b.ny + NK_BY - 1) / NK_BY, b.nz); }
static dim3 nk_level_grid(int nslots, int maxnx, int maxny, int maxnz)
{
    return dim3((maxnx + NK_BX - 1) / NK_BX, (maxny + NK_BY - 1) / NK_BY, maxnz * nslots);
}

void launch_closures_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const KParams& kp, hipStream_t s, int* floored)
{
    LEVEL_SPLIT(nslots, maxnz + 4, launch_closures_level(tab + s0_, n_, maxnx, maxny, maxnz, kp, s, floored));
    if (nslots <= 0) return;
    if (floored)
        hipLaunchKernelGGL(k_closures_level<true>, nk_level_grid(nslots, maxnx, maxny, maxnz), dim3(NK_BX, NK_BY, 1), 0, s, tab, maxnz, kp, floored);
    else
        hipLaunchKernelGGL(k_closures_level<false>, nk_level_grid(nslots, maxnx, maxny, maxnz), dim3(NK_BX, NK_BY, 1), 0, s, tab, maxnz, kp,
                           (int*)nullptr);
}
void launch_set_w_closures_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const double* vec, double turbFloor,
                                 const KParams& kp, int* floored, hipStream_t s)
{
    LEVEL_SPLIT(nslots, maxnz + 4, launch_set_w_closures_level(tab + s0_, n_, maxnx, maxny, maxnz, vec, turbFloor, kp, floored, s));
    if (nslots <= 0) return;
    hipLaunchKernelGGL(k_set_w_closures_level, nk_level_grid(nslots, maxnx, maxny, maxnz), dim3(NK_BX, NK_BY, 1), 0, s, tab, maxnz, vec, turbFloor,
                       kp, floored);
}
void launch_set_w_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, const double* vec, double turbFloor, hipStream_t s)
{
    LEVEL_SPLIT(nslots, maxnz + 4, launch_set_w_level(tab + s0_, n_, maxnx, maxny, maxnz, vec, turbFloor, s));
    if (nslots <= 0) return;
    hipLaunchKernelGGL(k_set_w_level, nk_level_grid(nslots, maxnx, maxny, maxnz), dim3(NK_BX, NK_BY, 1), 0, s, tab, maxnz, vec, turbFloor);
}
void launch_get_r_level(const BlkView* tab, int nslots, int maxnx, int maxny, int maxnz, double* vec, double turbScale, double* sums, hipStream_t s)
{
    LEVEL_SPLIT(nslots, maxnz + 4, launch_get_r_level(tab + s0_, n_, maxnx, maxny, maxnz, vec, turbScale, sums, s));
    if (nslots <= 0) return;
    hipLaunchKernelGGL(k_get_r_level, nk_level_grid(nslots, maxnx, maxny, maxnz), dim3(NK_BX, NK_BY, 1), 0, s, tab, maxnz, vec, turbScale, sums);
}

void launch_set_w(const BlkView& b, const double* vec, double turbFloor, hipStream_t s)
{
    hipLaunchKernelGGL(k_set_w, nk_grid(b), dim3(NK_BX, NK_BY, 1), 0, s, b, vec, turbFloor);
}

void launch_get_r(const BlkView& b, double* vec, double turbScale, double* sums, hipStream_t s)
{
    hipLaunchKernelGGL(k_get_r, nk_grid(b), dim3(NK_BX, NK_BY, 1), 0, s, b, vec, turbScale, sums);
}

void launch_closures(const BlkView& b, const KParams& kp, hipStream_t s)
{
    hipLaunchKernelGGL(k_closures, nk_grid(b), dim3(NK_BX, NK_BY, 1), 0, s, b, kp);
}
