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

__global__ void k_halo_copy(const BlkView* __restrict__ tab, const int* __restrict__ donorBlk,
                            const long* __restrict__ donorOff, const int* __restrict__ haloBlk,
                            const long* __restrict__ haloOff, int n, unsigned mask)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const BlkView& db = tab[donorBlk[t]];
    const BlkView& hb = tab[haloBlk[t]];
    const long dof = donorOff[t], hof = haloOff[t];
    // every value is requested before the first is stored: a store in between would order the loads behind it (the compiler cannot
    // know that halos and donors never overlap) and leave ONE 8-byte load in flight per lane -- 2.3 TB/s of payload on the 8-block mesh
    double val[HALO_NVAR];
#pragma unroll
    for (int v = 0; v < HALO_NVAR; ++v)
        if (mask & (1u << v)) val[v] = halo_var(db, v)[dof];
#pragma unroll
    for (int v = 0; v < HALO_NVAR; ++v)
        if (mask & (1u << v)) halo_var(hb, v)[hof] = val[v];
}

__global__ void k_halo_pack(const BlkView* __restrict__ tab, const int* __restrict__ blk, const long* __restrict__ off,
                            int n, unsigned mask, double* __restrict__ buf)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const BlkView& b = tab[blk[t]];
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

__global__ void k_halo_unpack(const BlkView* __restrict__ tab, const int* __restrict__ blk, const long* __restrict__ off,
                              int n, unsigned mask, const double* __restrict__ buf)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const BlkView& b = tab[blk[t]];
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

void launch_halo_copy(const BlkView* tab, const int* donorBlk, const long* donorOff, const int* haloBlk, const long* haloOff,
                      int n, unsigned mask, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_halo_copy, dim3((n + 255) / 256), dim3(256), 0, s, tab, donorBlk, donorOff, haloBlk, haloOff, n, mask);
}

void launch_halo_pack(const BlkView* tab, const int* blk, const long* off, int n, unsigned mask, double* buf, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_halo_pack, dim3((n + 255) / 256), dim3(256), 0, s, tab, blk, off, n, mask, buf);
}

void launch_halo_unpack(const BlkView* tab, const int* blk, const long* off, int n, unsigned mask, const double* buf,
                        hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_halo_unpack, dim3((n + 255) / 256), dim3(256), 0, s, tab, blk, off, n, mask, buf);
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
