// Ghost-cell (halo) exchange kernels: same-GPU block-to-block copy and the
// pack / unpack of inter-GPU messages.
//
// Reference semantics: haloExchange::whalo1to1RealGeneric
// (src/utils/haloExchange.F90:553-719): per neighbour rank pack nVar values per
// cell -> isend ; irecv ; same-rank copies (:657-678) ; waitany + unpack.  The
// index lists are the reference's own (communication.F90 commType /
// internalCommType), converted once to linear offsets of the HBM box layout.
// Message layout here is variable-major (buf[v*n + j]) so that both the pack
// store and the unpack load are coalesced; the reference packs cell-major.
#include "internal.h"

// variable selector: bit l (0..5) = w(:,:,:,l+1), bit 8 = p, bit 9 = rlv, bit 10 = rev, bits 11..13 = x(:,:,:,1:3)
// (node coordinates: the entries of a NODE pattern, exchangeCoor)
#define HALO_NVAR 14
__device__ __forceinline__ double* halo_var(const BlkView& b, int v)
{
    if (v < 8) return b.w + (long)v * b.nbox;
    if (v == 8) return b.p;
    if (v == 9) return b.rlv;
    if (v == 10) return b.rev;
    return b.x + (long)(v - 11) * b.nbox;
}

// base pointers of the exchangeable arrays of one block, read from the table BEFORE any data moves (a table load between the data
// loads is a dependent round trip of its own, and behind a store the compiler must assume the table changed)
struct HaloBase { double *w, *p, *rlv, *rev, *x; long nbox; };
__device__ __forceinline__ HaloBase halo_base(const BlkView& b)
{
    HaloBase h;
    h.w = b.w; h.p = b.p; h.rlv = b.rlv; h.rev = b.rev; h.x = b.x; h.nbox = b.nbox;
    return h;
}
__device__ __forceinline__ double* halo_var(const HaloBase& b, int v)
{
    if (v < 8) return b.w + (long)v * b.nbox;
    if (v == 8) return b.p;
    if (v == 9) return b.rlv;
    if (v == 10) return b.rev;
    return b.x + (long)(v - 11) * b.nbox;
}
// every value is requested before the first is stored: a store in between would order the loads behind it (the compiler cannot
// know that halos and donors never overlap) and leave ONE 8-byte load in flight per lane -- 2.3 TB/s of payload on the 8-block mesh
// CM != 0: the variable set at compile time (the common exchanges: no branch between the loads -- with a run-time mask the compiler
// puts a full wait in front of every conditional load and the nine loads of a halo cell go one after the other)
template <unsigned CM>
__device__ __forceinline__ void halo_move(const HaloBase& db, const HaloBase& hb, long dof, long hof, unsigned rmask)
{
    const unsigned mask = CM ? CM : rmask;
    double val[HALO_NVAR];
#pragma unroll
    for (int v = 0; v < HALO_NVAR; ++v)
        if (mask & (1u << v)) val[v] = halo_var(db, v)[dof];
#pragma unroll
    for (int v = 0; v < HALO_NVAR; ++v)
        if (mask & (1u << v)) halo_var(hb, v)[hof] = val[v];
}

template <unsigned CM>
__global__ void k_halo_copy(const BlkView* __restrict__ tab, const int* __restrict__ donorBlk,
                            const long* __restrict__ donorOff, const int* __restrict__ haloBlk,
                            const long* __restrict__ haloOff, int n, unsigned mask)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int dbi = donorBlk[t], hbi = haloBlk[t];
    const long dof = donorOff[t], hof = haloOff[t];
#ifndef HOSTSIM
    // The lists are sorted by destination: almost every wavefront copies from ONE block into ONE block.  Then the two table entries
    // are read once per wave through the scalar unit instead of a dozen 8-byte table loads per lane (round 5: on 343 blocks of 32^3
    // cells the copies of 4.76 M halo cells were 0.43 of the 3.45 ms step; the table pointers were loaded one by one between the data
    // loads, each waited for)
    const int d0 = __builtin_amdgcn_readfirstlane(dbi), h0 = __builtin_amdgcn_readfirstlane(hbi);
    if (__builtin_amdgcn_ballot_w64(dbi != d0 || hbi != h0) == 0) {
        const HaloBase db = halo_base(tab[d0]), hb = halo_base(tab[h0]);
        halo_move<CM>(db, hb, dof, hof, mask);
        return;
    }
#endif
    const HaloBase db = halo_base(tab[dbi]), hb = halo_base(tab[hbi]);
    halo_move<CM>(db, hb, dof, hof, mask);
}

template <unsigned CM>
__global__ void k_halo_pack(const BlkView* __restrict__ tab, const int* __restrict__ blk, const long* __restrict__ off,
                            int n, unsigned rmask, double* __restrict__ buf)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const unsigned mask = CM ? CM : rmask;
    const HaloBase b = halo_base(tab[blk[t]]);
    const long o = off[t];
    double val[HALO_NVAR];
#pragma unroll
    for (int v = 0; v < HALO_NVAR; ++v)
        if (mask & (1u << v)) val[v] = halo_var(b, v)[o];
    int q = 0;
#pragma unroll
    for (int v = 0; v < HALO_NVAR; ++v)
        if (mask & (1u << v)) {
            buf[(long)q * n + t] = val[v];
            ++q;
        }
}

template <unsigned CM>
__global__ void k_halo_unpack(const BlkView* __restrict__ tab, const int* __restrict__ blk, const long* __restrict__ off,
                              int n, unsigned rmask, const double* __restrict__ buf)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const unsigned mask = CM ? CM : rmask;
    const HaloBase b = halo_base(tab[blk[t]]);
    const long o = off[t];
    double val[HALO_NVAR];
    int q = 0;
#pragma unroll
    for (int v = 0; v < HALO_NVAR; ++v)
        if (mask & (1u << v)) {
            val[v] = buf[(long)q * n + t];
            ++q;
        }
#pragma unroll
    for (int v = 0; v < HALO_NVAR; ++v)
        if (mask & (1u << v)) halo_var(b, v)[o] = val[v];
}

// periodic transformations on the receiving side (haloExchange.F90:487-551 velocities, :2644-2712 coordinates)
struct Rot3 { double m[9]; double c[3], t[3]; };    // m column-major

__global__ void k_periodic_velocity(const BlkView* __restrict__ tab, const int* __restrict__ blk, const long* __restrict__ off, int n,
                                    Rot3 r)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const BlkView& b = tab[blk[t]];
    const long o = off[t], nb = b.nbox;
    const double vx = b.w[o + nb], vy = b.w[o + 2 * nb], vz = b.w[o + 3 * nb];
    b.w[o + nb] = r.m[0] * vx + r.m[3] * vy + r.m[6] * vz;
    b.w[o + 2 * nb] = r.m[1] * vx + r.m[4] * vy + r.m[7] * vz;
    b.w[o + 3 * nb] = r.m[2] * vx + r.m[5] * vy + r.m[8] * vz;
}

__global__ void k_periodic_coor(const BlkView* __restrict__ tab, const int* __restrict__ blk, const long* __restrict__ off, int n, Rot3 r)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const BlkView& b = tab[blk[t]];
    const long o = off[t], nb = b.nbox;
    const double dx = b.x[o] - r.c[0], dy = b.x[o + nb] - r.c[1], dz = b.x[o + 2 * nb] - r.c[2];
    b.x[o] = r.m[0] * dx + r.m[3] * dy + r.m[6] * dz + r.t[0];
    b.x[o + nb] = r.m[1] * dx + r.m[4] * dy + r.m[7] * dz + r.t[1];
    b.x[o + 2 * nb] = r.m[2] * dx + r.m[5] * dy + r.m[8] * dz + r.t[2];
}

// coor: node pattern (translation already holds periodicData%translation + rotCenter)
void launch_periodic(const BlkView* tab, const int* blk, const long* off, int n, const double rotMatrix[9], const double rotCenter[3],
                     const double translation[3], int coor, hipStream_t s)
{
    if (n <= 0) return;
    Rot3 r;
    for (int q = 0; q < 9; ++q) r.m[q] = rotMatrix[q];
    for (int q = 0; q < 3; ++q) { r.c[q] = rotCenter[q]; r.t[q] = translation[q] + rotCenter[q]; }
    if (coor) hipLaunchKernelGGL(k_periodic_coor, dim3((n + 255) / 256), dim3(256), 0, s, tab, blk, off, n, r);
    else hipLaunchKernelGGL(k_periodic_velocity, dim3((n + 255) / 256), dim3(256), 0, s, tab, blk, off, n, r);
}

// the variable sets of the common exchanges get kernels of their own (mask at compile time); anything else takes the run-time form
#define HALO_MASK_RANS 0x73Fu      // w(1:6), p, rlv, rev: whalo2 of a RANS level
#define HALO_MASK_NS 0x71Fu        // w(1:5), p, rlv, rev: the mean-flow part of a RANS level
#define HALO_MASK_LAM 0x31Fu       // w(1:5), p, rlv: laminar
#define HALO_MASK_EULER 0x11Fu     // w(1:5), p
#define HALO_MASK_P 0x100u         // the early pressure exchange
#define HALO_MASK_TURB 0x620u      // w(6), rlv, rev: whalo2(nt1:nt2) of the SA solve
#define HALO_DISPATCH(KERNEL, ...)                                                                                         \
    switch (mask) {                                                                                                        \
    case HALO_MASK_RANS: hipLaunchKernelGGL((KERNEL<HALO_MASK_RANS>), dim3((n + 255) / 256), dim3(256), 0, s, __VA_ARGS__); break;   \
    case HALO_MASK_NS: hipLaunchKernelGGL((KERNEL<HALO_MASK_NS>), dim3((n + 255) / 256), dim3(256), 0, s, __VA_ARGS__); break;       \
    case HALO_MASK_LAM: hipLaunchKernelGGL((KERNEL<HALO_MASK_LAM>), dim3((n + 255) / 256), dim3(256), 0, s, __VA_ARGS__); break;     \
    case HALO_MASK_EULER: hipLaunchKernelGGL((KERNEL<HALO_MASK_EULER>), dim3((n + 255) / 256), dim3(256), 0, s, __VA_ARGS__); break; \
    case HALO_MASK_P: hipLaunchKernelGGL((KERNEL<HALO_MASK_P>), dim3((n + 255) / 256), dim3(256), 0, s, __VA_ARGS__); break;         \
    case HALO_MASK_TURB: hipLaunchKernelGGL((KERNEL<HALO_MASK_TURB>), dim3((n + 255) / 256), dim3(256), 0, s, __VA_ARGS__); break;   \
    default: hipLaunchKernelGGL((KERNEL<0u>), dim3((n + 255) / 256), dim3(256), 0, s, __VA_ARGS__); break;                          \
    }

void launch_halo_copy(const BlkView* tab, const int* donorBlk, const long* donorOff, const int* haloBlk, const long* haloOff,
                      int n, unsigned mask, hipStream_t s)
{
    if (n <= 0) return;
    HALO_DISPATCH(k_halo_copy, tab, donorBlk, donorOff, haloBlk, haloOff, n, mask)
}

void launch_halo_pack(const BlkView* tab, const int* blk, const long* off, int n, unsigned mask, double* buf, hipStream_t s)
{
    if (n <= 0) return;
    HALO_DISPATCH(k_halo_pack, tab, blk, off, n, mask, buf)
}

void launch_halo_unpack(const BlkView* tab, const int* blk, const long* off, int n, unsigned mask, const double* buf,
                        hipStream_t s)
{
    if (n <= 0) return;
    HALO_DISPATCH(k_halo_unpack, tab, blk, off, n, mask, buf)
}

// sum over owned cells of (dw(:,l)/vol)^2 for l = 0..n-1  (solvers.F90:1538 monitoring sums)
__global__ void k_res_norms(BlkView b, int nvar, double* __restrict__ out)
{
    __shared__ double red[256];
    const int tid = threadIdx.x + blockDim.x * threadIdx.y;
    const int i = blockIdx.x * 64 + threadIdx.x + 2;
    const int j = blockIdx.y * 4 + threadIdx.y + 2;
    const int k = blockIdx.z + 2;
    const bool in = (i <= b.il && j <= b.jl);
    const long c = in ? b.idx(i, j, k) : 0;
    for (int l = 0; l < nvar; ++l) {
        double v = 0.0;
        if (in) {
            const double r = b.dw[c + l * b.nbox] / b.vol[c];
            v = r * r;
        }
        red[tid] = v;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) red[tid] += red[tid + st];
            __syncthreads();
        }
        if (tid == 0) atomicAdd(&out[l], red[0]);
        __syncthreads();
    }
}

void launch_res_norms(const BlkView& b, int nvar, double* out, hipStream_t s)
{
    dim3 blk(64, 4, 1);
    dim3 grd((b.nx + 63) / 64, (b.ny + 3) / 4, b.nz);
    hipLaunchKernelGGL(k_res_norms, grd, blk, 0, s, b, nvar, out);
}
