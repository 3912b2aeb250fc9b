// Calibration helpers run on the GPU box next to the profiles (test / measurement infrastructure, not product code):
//   copy8 / copy16 : streaming copies of a known byte count with 8-byte and 16-byte accesses per lane, so that the
//                    FETCH_SIZE / WRITE_SIZE counters of rocprofv3 can be calibrated for the access widths the flux
//                    kernels use (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own pattern");
//   probe          : accuracy of the raw v_rcp_f64 / v_rsq_f64 seeds and of rcp_nr / rsq_nr after 1 and 2 Newton steps
//                    (internal.h), which decides ADF_NR.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/pmc_calib.bin tools/pmc_calib.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void copy8(double* __restrict__ d, const double* __restrict__ s, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void copy16(double2* __restrict__ d, const double2* __restrict__ s, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
// five separate 8-byte streams read, one written (the access shape of the flux kernels: SoA components)
__global__ void read5w1(double* __restrict__ d, const double* __restrict__ s, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        d[i] = s[i] + s[i + n] + s[i + 2 * n] + s[i + 3 * n] + s[i + 4 * n];
}

// streaming copy as the guide measures it (MI355X_MICROARCH.md: 6.29 TB/s with a float4 copy): 16 bytes per lane, U independent
// loads in flight per thread (256 threads x U x 16 B = 32 KiB per workgroup at U = 8), exact grid, no grid-stride tail
template <int U>
__global__ __launch_bounds__(256) void copy16u(double2* __restrict__ d, const double2* __restrict__ s)
{
    const size_t base = ((size_t)blockIdx.x * U) * 256 + threadIdx.x;
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = s[base + (size_t)u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) d[base + (size_t)u * 256] = v[u];
}
// read-only stream (sum into one value per thread, stored only if nonzero-impossible): the read side alone
template <int U>
__global__ __launch_bounds__(256) void read16u(const double2* __restrict__ s, double* __restrict__ out)
{
    const size_t base = ((size_t)blockIdx.x * U) * 256 + threadIdx.x;
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) { const double2 v = s[base + (size_t)u * 256]; acc += v.x + v.y; }
    if (acc == 123.456) out[threadIdx.x] = acc;
}
// FP64 issue rate: 8 independent FMA chains per lane, N trips: 8 N v_fma_f64 per wave, nothing else in the loop but the counter
__global__ __launch_bounds__(256) void valu_fma(double* __restrict__ out, int n, double a, double b)
{
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < n; ++i) {
        x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
        x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b);
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
}

#include "../adflow_amd/csrc/internal.h"
__global__ void probe_pow(const double* __restrict__ x, double* __restrict__ y, double a, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = fast_powa(x[i], a);
}

template <int NR>
__global__ void probe(const double* __restrict__ x, double* __restrict__ rc, double* __restrict__ rs, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double b = x[i];
    double r = __builtin_amdgcn_rcp(b);
    for (int it = 0; it < NR; ++it) { const double e = __builtin_fma(-b, r, 1.0); r = __builtin_fma(r, e, r); }
    double y = __builtin_amdgcn_rsq(b);
    for (int it = 0; it < NR; ++it) { const double t = b * y; const double e = __builtin_fma(-t, y, 1.0); y = __builtin_fma(0.5 * y, e, y); }
    rc[i] = r; rs[i] = y;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)


// "streams": the access shape of the fused viscous march without its arithmetic -- workgroups of four waves, 80 KB of LDS each (two per
// CU), wave r reads the rows j0-1+r and j0+r of NA arrays per plane (8 bytes per lane, 64 lanes = one 512-byte piece per array and
// row), one wait per plane, 18 planes per chunk -- over two layouts of the SAME bytes:
//   layout 0: structure of arrays, component stride = the whole box (what the library holds: 12.6 MB between the pieces of a row)
//   layout 1: row-blocked, component stride = one row (the NA pieces of a row are contiguous: NA x 1408 bytes)
// Says whether the 4.1 TB/s the marches draw is the memory system's answer to forty scattered 512-byte streams per tile.
template <int NA, int NF, int NS, bool OWN = false>
__global__ __launch_bounds__(256, 2) void streams(const double* __restrict__ base, double* __restrict__ out, int layout, long nbox, int ldi, long ldk,
                                                  int ntx, int nty, int nch, int planes, double* __restrict__ wr)
{
    __shared__ double pad[10000];                       // 80 KB: two workgroups per CU, as k_visc_gf
    const int lane = threadIdx.x, r = threadIdx.y;
    int t = blockIdx.x;
    if (layout >= 2) {      // the library's launch order: within every round of 512 resident workgroups XCD x (= blockIdx % 8) takes the
        const int q = t / 512, pr = t % 512;      // x-th contiguous 64 of the round's tiles (j-neighbours, which share rows, on one L2)
        t = q * 512 + (pr % 8) * 64 + pr / 8;
        if (t >= (int)gridDim.x) return;
    }
    const int bx = t % ntx; t /= ntx;
    const int by = t % nty; t /= nty;
    const int ch = t % nch; const int blk = t / nch;
    const int i = bx * 60 + lane, j = by * 3 + r + 1, k0 = ch * planes + 1;
    const long cstride = (layout & 1) == 0 ? nbox : (long)ldi;                 // doubles between components
    const long rowA = (layout & 1) == 0 ? (long)j * ldi : (long)j * ldi * (NA + NF);
    const long rowB = (layout & 1) == 0 ? (long)(j + 1) * ldi : (long)(j + 1) * ldi * (NA + NF);
    const long kst = (layout & 1) == 0 ? ldk : ldk * (NA + NF);
    const double* b = base + (long)blk * nbox * (NA + NF);
    double acc = 0.0;
    for (int k = k0; k < k0 + planes + 2; ++k) {
        double v[2 * NA];
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            v[2 * a] = b[(long)k * kst + rowA + a * cstride + i];
            v[2 * a + 1] = (!OWN || r == 3) ? b[(long)k * kst + rowB + a * cstride + i] : 0.0;   // OWN: the row above through LDS (not modelled)
        }
#pragma unroll
        for (int a = 0; a < 2 * NA; ++a) acc += v[a];
        // the face part: NF more arrays of the own row at plane k-1 (behind the gradient arrays in the buffer), six re-reads of gradient
        // arrays of that plane (L2), NS stores
        double f[NF + 6];
#pragma unroll
        for (int a = 0; a < NF; ++a) f[a] = b[(long)(k - 1) * kst + rowA + (NA + a) * cstride + i];
#pragma unroll
        for (int a = 0; a < 6; ++a) f[NF + a] = b[(long)(k - 1) * kst + rowA + a * cstride + i];
        __syncthreads();
#pragma unroll
        for (int a = 0; a < NF + 6; ++a) acc += f[a];
        if (NS > 0 && r >= 1 && lane >= 2 && lane < 62) {
#pragma unroll
            for (int a = 0; a < NS; ++a) wr[(long)blk * nbox * NS + (long)k * ldk + (long)j * ldi + a * nbox + i] = acc;
        }
    }
    if (acc == 123.456) { pad[lane] = acc; out[lane] = pad[lane ^ 1]; }
}

// the same with the gradient part's loads requested ONE PLANE AHEAD (registers are free here): what a prefetching form of the march
// could reach at the same occupancy
template <int NA, int NF, int NS>
__global__ __launch_bounds__(256, 2) void streams_pf(const double* __restrict__ base, double* __restrict__ out, int layout, long nbox, int ldi, long ldk,
                                                     int ntx, int nty, int nch, int planes, double* __restrict__ wr)
{
    __shared__ double pad[10000];
    const int lane = threadIdx.x, r = threadIdx.y;
    int t = blockIdx.x;
    if (layout >= 2) {      // the library's launch order: within every round of 512 resident workgroups XCD x (= blockIdx % 8) takes the
        const int q = t / 512, pr = t % 512;      // x-th contiguous 64 of the round's tiles (j-neighbours, which share rows, on one L2)
        t = q * 512 + (pr % 8) * 64 + pr / 8;
        if (t >= (int)gridDim.x) return;
    }
    const int bx = t % ntx; t /= ntx;
    const int by = t % nty; t /= nty;
    const int ch = t % nch; const int blk = t / nch;
    const int i = bx * 60 + lane, j = by * 3 + r + 1, k0 = ch * planes + 1;
    const long cstride = (layout & 1) == 0 ? nbox : (long)ldi;
    const long rowA = (layout & 1) == 0 ? (long)j * ldi : (long)j * ldi * (NA + NF);
    const long rowB = (layout & 1) == 0 ? (long)(j + 1) * ldi : (long)(j + 1) * ldi * (NA + NF);
    const long kst = (layout & 1) == 0 ? ldk : ldk * (NA + NF);
    const double* b = base + (long)blk * nbox * (NA + NF);
    double acc = 0.0;
    double v[2 * NA], w[2 * NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) { v[2 * a] = b[(long)k0 * kst + rowA + a * cstride + i]; v[2 * a + 1] = b[(long)k0 * kst + rowB + a * cstride + i]; }
    for (int k = k0; k < k0 + planes + 2; ++k) {
#pragma unroll
        for (int a = 0; a < NA; ++a) { w[2 * a] = b[(long)(k + 1) * kst + rowA + a * cstride + i]; w[2 * a + 1] = b[(long)(k + 1) * kst + rowB + a * cstride + i]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int a = 0; a < 2 * NA; ++a) acc += v[a];
        double f[NF + 6];
#pragma unroll
        for (int a = 0; a < NF; ++a) f[a] = b[(long)(k - 1) * kst + rowA + (NA + a) * cstride + i];
#pragma unroll
        for (int a = 0; a < 6; ++a) f[NF + a] = b[(long)(k - 1) * kst + rowA + a * cstride + i];
        __syncthreads();
#pragma unroll
        for (int a = 0; a < NF + 6; ++a) acc += f[a];
        if (NS > 0 && r >= 1 && lane >= 2 && lane < 62) {
#pragma unroll
            for (int a = 0; a < NS; ++a) wr[(long)blk * nbox * NS + (long)k * ldk + (long)j * ldi + a * nbox + i] = acc;
        }
#pragma unroll
        for (int a = 0; a < 2 * NA; ++a) v[a] = w[a];
    }
    if (acc == 123.456) { pad[lane] = acc; out[lane] = pad[lane ^ 1]; }
}

int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "all";
    if (!strcmp(mode, "copy") || !strcmp(mode, "all")) {
        const size_t n = (size_t)1 << 27;          // 128 Mi doubles = 1 GiB per stream: far beyond the 256 MiB Infinity Cache
        double *s, *d;
        CK(hipMalloc(&s, 5 * n * sizeof(double) / 4 + n * sizeof(double)));   // read5w1 reads 5 streams of n/4
        CK(hipMalloc(&d, n * sizeof(double)));
        CK(hipMemset(s, 0, 5 * n * sizeof(double) / 4 + n * sizeof(double)));
        for (int it = 0; it < 5; ++it) {
            hipLaunchKernelGGL(copy8, dim3(8192), dim3(256), 0, 0, d, s, n);
            hipLaunchKernelGGL(copy16, dim3(8192), dim3(256), 0, 0, (double2*)d, (const double2*)s, n / 2);
            hipLaunchKernelGGL(read5w1, dim3(8192), dim3(256), 0, 0, d, s, n / 4);
        }
        CK(hipDeviceSynchronize());
        printf("copy8: reads %zu B writes %zu B per launch; copy16: the same; read5w1: reads %zu B writes %zu B\n", n * 8, n * 8,
               5 * (n / 4) * 8, (n / 4) * 8);
    }
    if (!strcmp(mode, "pow")) {
        // accuracy of fast_powa (internal.h) against the host's pow over the range the spectral radii live in
        const size_t n = 1 << 22;
        std::vector<double> hx(n), hy(n);
        srand(11);
        for (size_t i = 0; i < n; ++i) hx[i] = exp(((double)rand() / RAND_MAX) * 80.0 - 58.0);   // 1e-25 .. 3e9
        double *x, *y;
        CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8));
        CK(hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice));
        for (double a : {0.67, 2.0 / 3.0, 0.5, 1.0, 0.25}) {
            hipLaunchKernelGGL(probe_pow, dim3((n + 255) / 256), dim3(256), 0, 0, x, y, a, n);
            CK(hipMemcpy(hy.data(), y, n * 8, hipMemcpyDeviceToHost));
            double er = 0;
            for (size_t i = 0; i < n; ++i) er = fmax(er, fabs(hy[i] / pow(hx[i], a) - 1.0));
            printf("fast_powa(x, %.4f): max rel err %.3e over x in 1e-25 .. 3e9\n", a, er);
        }
    }
    if (!strcmp(mode, "bw")) {
        // achievable HBM bandwidth of the three access shapes (hipEvent timing, 10 launches each): what "peak" means in practice
        const size_t n = (size_t)1 << 27;
        double *s, *d;
        CK(hipMalloc(&s, 5 * n * sizeof(double) / 4 + n * sizeof(double)));
        CK(hipMalloc(&d, n * sizeof(double)));
        CK(hipMemset(s, 0, 5 * n * sizeof(double) / 4 + n * sizeof(double)));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int which = 0; which < 3; ++which) {
            for (int it = -2; it < 10; ++it) {
                if (it == 0) CK(hipEventRecord(e0, 0));
                if (which == 0) hipLaunchKernelGGL(copy8, dim3(8192), dim3(256), 0, 0, d, s, n);
                if (which == 1) hipLaunchKernelGGL(copy16, dim3(8192), dim3(256), 0, 0, (double2*)d, (const double2*)s, n / 2);
                if (which == 2) hipLaunchKernelGGL(read5w1, dim3(8192), dim3(256), 0, 0, d, s, n / 4);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (which < 2) ? 16.0 * n : 6.0 * (n / 4) * 8.0;
            printf("%s: %.3f ms per launch, %.0f GB/s (read + write)\n", which == 0 ? "copy8" : which == 1 ? "copy16" : "read5w1", ms / 10.0,
                   bytes / (ms / 10.0 * 1e-3) / 1e9);
        }
    }
    if (!strcmp(mode, "bw2")) {
        // (a) the guide's streaming ceiling with our own kernel: 16 B per lane, 8 loads in flight per thread, exact grid, > 256 MiB
        //     working set (HBM) and a 64 MiB one (Infinity Cache: FETCH_SIZE counts these requests too, they are not HBM traffic);
        // (b) the sustained FP64 issue rate: v_fma_f64 per second per SIMD -> the clock the issue roofline of bench.py is priced at
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipDeviceProp_t prop;
        CK(hipGetDeviceProperties(&prop, 0));
        printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
        for (int ws = 0; ws < 2; ++ws) {
            const size_t n2 = ws == 0 ? ((size_t)1 << 26) : ((size_t)1 << 21);   // double2 elements: 1 GiB / 32 MiB per array
            double2 *s, *d; double* o;
            CK(hipMalloc(&s, n2 * 16)); CK(hipMalloc(&d, n2 * 16)); CK(hipMalloc(&o, 4096));
            CK(hipMemset(s, 0, n2 * 16));
            const int reps = ws == 0 ? 10 : 200;
            for (int which = 0; which < 2; ++which) {
                // best of three rounds: the first timed round of a fresh process has been seen at a third of the rate (r06_fin3)
                float best = 1e30f;
                for (int rnd = 0; rnd < 3; ++rnd) {
                    for (int it = -2; it < reps; ++it) {
                        if (it == 0) CK(hipEventRecord(e0, 0));
                        if (which == 0) hipLaunchKernelGGL(copy16u<8>, dim3((unsigned)(n2 / (8 * 256))), dim3(256), 0, 0, d, s);
                        else hipLaunchKernelGGL(read16u<8>, dim3((unsigned)(n2 / (8 * 256))), dim3(256), 0, 0, s, o);
                    }
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                const double bytes = (which == 0 ? 32.0 : 16.0) * n2;
                printf(", \"%s_%s_gbs\": %.0f", which == 0 ? "copy16u8" : "read16u8", ws == 0 ? "1gib" : "32mib", bytes / (best / reps * 1e-3) / 1e9);
            }
            CK(hipFree(s)); CK(hipFree(d)); CK(hipFree(o));
        }
        {
            double* o;
            const int wgs = prop.multiProcessorCount * 8, trips = 1 << 16;      // 8 workgroups x 4 waves per CU = 8 waves per SIMD
            CK(hipMalloc(&o, (size_t)wgs * 256 * 8));
            for (int it = -1; it < 3; ++it) {
                if (it == 0) CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(valu_fma, dim3(wgs), dim3(256), 0, 0, o, trips, 1.0000001, 1.e-9);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double wave_insts = (double)wgs * 4.0 * 8.0 * trips;          // v_fma_f64 issued by all waves of one launch
            const double per_simd = wave_insts / (ms / 3.0 * 1e-3) / (prop.multiProcessorCount * 4.0);
            printf(", \"fma64_wave_insts_per_s_per_simd\": %.4e, \"issue_clock_ghz_at_4_cycles\": %.3f}\n", per_simd, per_simd * 4.0 / 1e9);
        }
    }
    if (!strcmp(mode, "streams")) {
        // 8 blocks of 160x128x64 cells (box 176 x 132 x 68), tiles of 60 columns x 3 rows, chunks of 16 planes
        const int ldi = 176, nj = 132, nk = 68, nblk = 8, ntx = 3, nty = 43, planes = 16, nch = 4;
        const long ldk = (long)ldi * nj, nbox = ldk * nk;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        double *buf, *o;
        const int NAmax = 27;
        double* wr;
        CK(hipMalloc(&buf, (size_t)nbox * NAmax * nblk * 8 + (1 << 20))); CK(hipMalloc(&o, 4096));
        CK(hipMalloc(&wr, (size_t)nbox * 4 * nblk * 8 + (1 << 20)));
        CK(hipMemset(buf, 0, (size_t)nbox * NAmax * nblk * 8));
        printf("{\"what\": \"4-wave workgroups, 2 per CU, two rows x NA arrays x 512 B per wave and plane, no arithmetic\"");
        for (int layout = 0; layout < 4; ++layout) {      // 0 / 1: SoA / row-blocked in launch order; 2 / 3: the same in the XCD-aware order
            const int reps = 20;
            for (int it = -2; it < reps; ++it) {
                if (it == 0) CK(hipEventRecord(e0, 0));
                if (argc > 2 && !strcmp(argv[2], "grad"))
                    hipLaunchKernelGGL((streams<17, 0, 0>), dim3(ntx * nty * nch * nblk), dim3(64, 4), 0, 0, buf, o, layout, nbox * 27 / 17 * 0 + nbox, ldi, ldk, ntx, nty, nch, planes, wr);
                else if (argc > 2 && !strcmp(argv[2], "own"))
                    hipLaunchKernelGGL((streams<17, 10, 4, true>), dim3(ntx * nty * nch * nblk), dim3(64, 4), 0, 0, buf, o, layout, nbox, ldi, ldk, ntx, nty, nch, planes, wr);
                else if (argc > 2 && !strcmp(argv[2], "pf"))
                    hipLaunchKernelGGL((streams_pf<17, 10, 4>), dim3(ntx * nty * nch * nblk), dim3(64, 4), 0, 0, buf, o, layout, nbox, ldi, ldk, ntx, nty, nch, planes, wr);
                else
                    hipLaunchKernelGGL((streams<17, 10, 4>), dim3(ntx * nty * nch * nblk), dim3(64, 4), 0, 0, buf, o, layout, nbox, ldi, ldk, ntx, nty, nch, planes, wr);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            // bytes requested by the waves (rows shared by the waves of a workgroup counted once: 5 rows per tile and plane)
            const double req = (double)ntx * nty * nch * nblk * (planes + 2) * 5.0 * 17 * 512.0;
            printf(", \"layout%d_ms\": %.4f, \"layout%d_distinct_gbs\": %.0f", layout, ms / reps, layout, req / (ms / reps * 1e-3) / 1e9);
        }
        printf("}\n");
        CK(hipFree(buf)); CK(hipFree(o));
    }
    if (!strcmp(mode, "probe") || !strcmp(mode, "all")) {
        const size_t n = 1 << 22;
        std::vector<double> hx(n), hr(n), hs(n);
        srand(7);
        for (size_t i = 0; i < n; ++i) hx[i] = exp(((double)rand() / RAND_MAX) * 92.0 - 46.0);   // 1e-20 .. 1e20
        double *x, *rc, *rs;
        CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&rc, n * 8)); CK(hipMalloc(&rs, n * 8));
        CK(hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice));
        for (int nr = 0; nr <= 2; ++nr) {
            if (nr == 0) hipLaunchKernelGGL(probe<0>, dim3((n + 255) / 256), dim3(256), 0, 0, x, rc, rs, n);
            if (nr == 1) hipLaunchKernelGGL(probe<1>, dim3((n + 255) / 256), dim3(256), 0, 0, x, rc, rs, n);
            if (nr == 2) hipLaunchKernelGGL(probe<2>, dim3((n + 255) / 256), dim3(256), 0, 0, x, rc, rs, n);
            CK(hipMemcpy(hr.data(), rc, n * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hs.data(), rs, n * 8, hipMemcpyDeviceToHost));
            double er = 0, es = 0;
            for (size_t i = 0; i < n; ++i) {
                er = fmax(er, fabs(hr[i] * hx[i] - 1.0));
                es = fmax(es, fabs(hs[i] * sqrt(hx[i]) - 1.0));
            }
            printf("NR=%d  max rel err  rcp %.3e   rsq %.3e\n", nr, er, es);
        }
    }
    return 0;
}
