// Coloured finite-difference Jacobian blocks of the residual on the device (second half).
//
// Reference semantics:
//   adjointUtils::setupStateResidualMatrix (useAD = F)   src/adjoint/adjointUtils.F90:7-715
//     colourings setup_PC_coloring / setup_dRdw_euler_coloring / setup_dRdw_visc_coloring   :1089-1185
//     stencils                                                                              src/modules/stencils.f90
//   masterRoutines::block_res_state                       src/adjoint/masterRoutines.F90:1214-1283
//     computePressureSimple(.True.) 0..ib, computeLamViscosity(.True.) / computeEddyViscosity(.True.) 1..ie
//     (flowUtils.F90:867-930, 1201-1300, turbUtils.F90:581-655), boundary conditions, residual core, resScale
//     (adjointExtra.F90:601-636)
//
// The sweep: for every colour and every state variable l, w(l) of ALL cells of that colour (halos included: the columns of the
// neighbouring blocks' cells) is raised by delta, the residual is evaluated, and every owned cell stores, for each stencil
// offset s whose source cell  row - s  has that colour,  d dw(row, :) / d w(row - s, l) = (dw - dw_ref) / delta.
// A valid colouring gives every cell of a stencil a different colour, so one residual evaluation fills one column of every block
// of the sparse matrix.  Layout of the result: component ((s * nState + l) * nState + ll) of a box array, i.e. the host sees
// (nx, ny, nz, nState, nState, nStencil) column-major = blk(ll, l) of the reference for every (row, s).
// These kernels are pointwise and run once per colour and variable next to a full residual evaluation: HBM-bound, not tuned.
#include "internal.h"

#define JC_BX 64
#define JC_BY 4

__device__ __forceinline__ int jc_colour(const JacSpec& J, int i, int j, int k)
{
    return (J.ca * (i % J.cm) + J.cb * (j % J.cm) + J.cc * (k % J.cm)) % J.cn;
}


__device__ __forceinline__ void closures_halo_at(const BlkView& b, const KParams& kp, int i, int j, int k, long c, const double wv[6]);

// w <- wref (+ delta on component l of the cells of colour `col`); col < 0: plain restore.  CLOS: also the closures of
// block_res_state from the new state (pressure on 0..ib, laminar / eddy viscosity on 1..ie) -- one pass instead of two
// onlyL: the other components hold wref already (the sweep of ONE state variable over the colours changes component l alone: every
// component is written at its first colour, afterwards 8 instead of 48 B per cell go out.  Taken for the preconditioner matrix only:
// api.hip adflow_gpu_fd_jacobian says why)
template <bool CLOS>
__global__ __launch_bounds__(JC_BX* JC_BY) void k_fd_state(BlkView b, const double* __restrict__ wref, int l, int col, JacSpec J,
                                                           double delta, KParams kp, int onlyL)
{
    const int i = blockIdx.x * JC_BX + threadIdx.x - 14;     // aligned rows (the box origin is shifted by ADF_PAD0)
    const int j = blockIdx.y * JC_BY + threadIdx.y;
    const int k = blockIdx.z;
    if (i < 0 || i > b.ib || j > b.jb) return;
    const long c = b.idx(i, j, k);
    const bool hit = (col >= 0) && (jc_colour(J, i, j, k) == col);
    double wv[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int m = 0; m < b.nw; ++m) {
        double v = wref[c + m * b.nbox];
        if (hit && m == l) v += delta;
        if (!onlyL || m == l) b.w[c + m * b.nbox] = v;
        wv[m] = v;
    }
    if (CLOS) closures_halo_at(b, kp, i, j, k, c, wv);
}

__global__ __launch_bounds__(JC_BX* JC_BY) void k_fd_copy(BlkView b, double* __restrict__ dst, const double* __restrict__ src, int ncomp)
{
    const int i = blockIdx.x * JC_BX + threadIdx.x - 14;
    const int j = blockIdx.y * JC_BY + threadIdx.y;
    const int k = blockIdx.z;
    if (i < 0 || i > b.ib || j > b.jb) return;
    const long c = b.idx(i, j, k);
    for (int m = 0; m < ncomp; ++m) dst[c + m * b.nbox] = src[c + m * b.nbox];
}

// pressure on 0..ib, laminar / eddy viscosity on 1..ie (the includeHalos = .True. forms used by block_res_state)
__device__ __forceinline__ void closures_halo_at(const BlkView& b, const KParams& kp, int i, int j, int k, long c, const double wv[6])
{
    const double rho = wv[0], u = wv[1], v = wv[2], w = wv[3];
    const double gm1 = kp.gammaConstant - 1.0;
    double p = gm1 * (wv[4] - 0.5 * rho * (u * u + v * v + w * w));
    p = fmax(p, 1.e-4 * kp.pInfCorr);
    b.p[c] = p;
    if (!kp.viscous || i < 1 || i > b.ie || j < 1 || j > b.je || k < 1 || k > b.ke) return;
    const double muSuth = kp.muSuthDim / kp.muRef, TSuth = kp.TSuthDim / kp.TRef, SSuth = kp.SSuthDim / kp.TRef;
    const double T = p / (kp.RGas * rho);
    const double tt = T / TSuth;
    const double rlv = muSuth * ((TSuth + SSuth) / (T + SSuth)) * (tt * sqrt(tt));
    b.rlv[c] = rlv;
    if (kp.eddyModel) {
        const double cv13 = kp.sa_cv1 * kp.sa_cv1 * kp.sa_cv1;
        const double rnuSA = wv[5] * rho;
        const double chi = rnuSA / rlv;
        const double chi3 = chi * chi * chi;
        b.rev[c] = chi3 / (chi3 + cv13) * rnuSA;
    }
}

__global__ __launch_bounds__(JC_BX* JC_BY) void k_closures_halo(BlkView b, KParams kp)
{
    const int i = blockIdx.x * JC_BX + threadIdx.x - 14;
    const int j = blockIdx.y * JC_BY + threadIdx.y;
    const int k = blockIdx.z;
    if (i < 0 || i > b.ib || j > b.jb) return;
    const long c = b.idx(i, j, k);
    double wv[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int m = 0; m < b.nw; ++m) wv[m] = b.w[c + m * b.nbox];
    closures_halo_at(b, kp, i, j, k, c, wv);
}

// owned cells: resScale (dw / volRef, turbulence * turbResScale) and store the reference residual (l = -1), or put it back into dw
// (l = -2, resetFDReference)
__global__ __launch_bounds__(JC_BX* JC_BY) void k_fd_extract(BlkView b, double* __restrict__ dwref, double* __restrict__ jac, int l, int col,
                                                             JacSpec J, double deltaInv, double turbResScale)
{
    const int i = blockIdx.x * JC_BX + threadIdx.x + 2;
    const int j = blockIdx.y * JC_BY + threadIdx.y + 2;
    const int k = blockIdx.z + 2;
    if (i > b.il || j > b.jl) return;
    const long c = b.idx(i, j, k);
    const long nb = b.nbox;
    const double ovol = 1.0 / b.volRef[c];
    double val[6];
    for (int m = 0; m < J.nState; ++m) {
        const int ll = J.lStart + m;
        val[m] = b.dw[c + ll * nb] * ovol * (ll >= 5 ? turbResScale : 1.0);
    }
    if (l == -1) {
        for (int m = 0; m < J.nState; ++m) dwref[c + m * nb] = val[m];
        return;
    }
    if (l == -2) {                    // resetFDReference: dw = the scaled reference residual
        for (int m = 0; m < J.nState; ++m) b.dw[c + (J.lStart + m) * nb] = dwref[c + m * nb];
        return;
    }
}

// ONE coloured evaluation, stored densely: snap[m] = resScale(dw)[m] on the owned cells (late round 5: the difference against the
// reference residual and the division by delta happen in k_fd_scatter, once per state variable and cell instead of once per
// evaluation -- 42 x 48 B per cell of dwref reads less in an assembly).
// The scatter into the stencil blocks happens once per state variable (k_fd_scatter): written per evaluation, every cell has ONE
// matching stencil entry and consecutive lanes hit different entries -- 1/nColour-dense 8-byte stores into nStencil separate
// streams ran at 0.3 TB/s (200 us per 1.3 M-cell block and evaluation, profiles/r02_ah)
__global__ __launch_bounds__(JC_BX* JC_BY) void k_fd_snap(BlkView b, double* __restrict__ snap, JacSpec J, double turbResScale)
{
    const int i = blockIdx.x * JC_BX + threadIdx.x + 2;
    const int j = blockIdx.y * JC_BY + threadIdx.y + 2;
    const int k = blockIdx.z + 2;
    if (i > b.il || j > b.jl) return;
    const long c = b.idx(i, j, k);
    const long nb = b.nbox;
    const double ovol = 1.0 / b.volRef[c];
    for (int m = 0; m < J.nState; ++m) {
        const int ll = J.lStart + m;
        const double val = b.dw[c + ll * nb] * ovol * (ll >= 5 ? turbResScale : 1.0);
        snap[c + m * nb] = val;
    }
}

// column l of every stencil block of the owned cells from the nColour snapshots of that state variable: entry s of the row cell
// takes the snapshot of the colour of its source cell row - offset(s).  Over its stencil a cell reads every snapshot once, but
// WHICH snapshot belongs to entry s differs from lane to lane (the colour of the source cell): read directly, one load touches
// nColour arrays with 1/nColour of the lanes each (0.50 ms per 1.3 M-cell block and variable, 8-byte gathers bound by the
// texture path).  So the snapshots of a cell pass through a lane-private LDS column: nColour coalesced loads, then nStencil
// coalesced stores that pick their value from the column.
// NC: capacity of the column (7 colours: 14 KB of LDS per workgroup; 35: 70 KB)
// dwref (finite differences): the snapshots hold the scaled residuals of the evaluations, the entry is (snapshot - dwref) / delta;
// NULL (forward mode): the snapshots hold the derivatives themselves
template <int NC>
__global__ __launch_bounds__(JC_BX* JC_BY) void k_fd_scatter(BlkView b, const double* __restrict__ snap, double* __restrict__ jac, int l,
                                                             JacSpec J, const double* __restrict__ dwref, double deltaInv)
{
    __shared__ double col[JC_BY][NC][JC_BX];
    const int i = blockIdx.x * JC_BX + threadIdx.x + 2;
    const int j = blockIdx.y * JC_BY + threadIdx.y + 2;
    const int k = blockIdx.z + 2;
    if (i > b.il || j > b.jl) return;
    const long c = b.idx(i, j, k);
    const long nb = b.nbox;
    const bool linear = (J.cm == J.cn);
    const int c0 = jc_colour(J, i, j, k);
    double (*mine)[JC_BX] = col[threadIdx.y];
    const int lane = threadIdx.x;
    for (int m = 0; m < J.nState; ++m) {
        if (dwref) {
            const double ref = dwref[c + m * nb];
            for (int d = 0; d < J.cn; ++d) mine[d][lane] = (snap[((long)d * J.nState + m) * nb + c] - ref) * deltaInv;
        } else
            for (int d = 0; d < J.cn; ++d) mine[d][lane] = snap[((long)d * J.nState + m) * nb + c];
        for (int s = 0; s < J.nStencil; ++s) {
            const int pi = i - J.st[s][0], pj = j - J.st[s][1], pk = k - J.st[s][2];      // the perturbed cell
            // (a column outside the box does not exist: its entry is written as zero -- every entry of every owned row is stored by the
            //  sweep, the 2 KB per cell of the matrix need no memset in front of an assembly)
            if (pi < 0 || pi > b.ib || pj < 0 || pj > b.jb || pk < 0 || pk > b.kb) {
                jac[c + ((long)(s * J.nState + (l - J.lStart)) * J.nState + m) * nb] = 0.0;
                continue;
            }
            int d;
            if (linear) {
                d = c0 - J.sc[s];
                if (d < 0) d += J.cn;
            } else
                d = jc_colour(J, pi, pj, pk);
            jac[c + ((long)(s * J.nState + (l - J.lStart)) * J.nState + m) * nb] = mine[d][lane];
        }
    }
}

// blocks of the owned cells of the planes k0 .. k0+nk-1 in the order of one MatSetValuesBlocked call per row cell:
// out(ll, l, s, i, j, k-k0), ll fastest.  A workgroup moves the nState x nState block of ONE stencil entry for 64 cells of a
// row: coalesced reads along i from the entry-major device storage, an LDS tile, runs of nState^2 doubles out.
#define JR_LD 37
__global__ __launch_bounds__(64) void k_jac_rows(BlkView b, const double* __restrict__ jac, double* __restrict__ out, int nState, int nStencil,
                                                 int k0)
{
    __shared__ double tile[64 * JR_LD];
    const int lane = threadIdx.x;
    const int i0 = blockIdx.x * 64 + 2;
    const int j = blockIdx.y + 2;
    const int s = blockIdx.z % nStencil, kk = blockIdx.z / nStencil;
    const int k = k0 + kk;
    const int nS2 = nState * nState;
    const long nb = b.nbox;
    const int i = i0 + lane;
    if (i <= b.il) {
        const long c = b.idx(i, j, k);
        for (int e = 0; e < nS2; ++e) tile[lane * JR_LD + e] = jac[c + ((long)s * nS2 + e) * nb];
    }
    __syncthreads();
    const int ncell = (b.il - i0 + 1 < 64) ? b.il - i0 + 1 : 64;
    const long row0 = (long)(i0 - 2) + (long)b.nx * ((j - 2) + (long)b.ny * kk);     // first cell of the tile in the slab's row order
    for (int t = lane; t < ncell * nS2; t += 64) {
        const int cell = t / nS2, e = t - cell * nS2;
        out[((row0 + cell) * nStencil + s) * nS2 + e] = tile[cell * JR_LD + e];
    }
}

void launch_jac_rows(const BlkView& b, const double* jac, double* out, int nState, int nStencil, int k0, int nk, hipStream_t s)
{
    hipLaunchKernelGGL(k_jac_rows, dim3((b.nx + 63) / 64, b.ny, nk * nStencil), dim3(64, 1, 1), 0, s, b, jac, out, nState, nStencil, k0);
}

static dim3 box_grid(const BlkView& b) { return dim3((b.ib + 15 + JC_BX) / JC_BX, (b.jb + JC_BY) / JC_BY, b.kb + 1); }
static dim3 own_grid(const BlkView& b) { return dim3((b.nx + JC_BX - 1) / JC_BX, (b.ny + JC_BY - 1) / JC_BY, b.nz); }

void launch_fd_state(const BlkView& b, const double* wref, int l, int col, const JacSpec& J, double delta, hipStream_t s)
{
    KParams kp = KParams();
    hipLaunchKernelGGL((k_fd_state<false>), box_grid(b), dim3(JC_BX, JC_BY, 1), 0, s, b, wref, l, col, J, delta, kp, 0);
}
// state of one coloured evaluation and the closures of block_res_state in one pass
void launch_fd_state_closures(const BlkView& b, const double* wref, int l, int col, const JacSpec& J, double delta, const KParams& kp,
                              hipStream_t s, bool onlyL)
{
    hipLaunchKernelGGL((k_fd_state<true>), box_grid(b), dim3(JC_BX, JC_BY, 1), 0, s, b, wref, l, col, J, delta, kp, onlyL ? 1 : 0);
}
void launch_fd_copy(const BlkView& b, double* dst, const double* src, int ncomp, hipStream_t s)
{
    hipLaunchKernelGGL(k_fd_copy, box_grid(b), dim3(JC_BX, JC_BY, 1), 0, s, b, dst, src, ncomp);
}
void launch_closures_halo(const BlkView& b, const KParams& kp, hipStream_t s)
{
    hipLaunchKernelGGL(k_closures_halo, box_grid(b), dim3(JC_BX, JC_BY, 1), 0, s, b, kp);
}
void launch_fd_extract(const BlkView& b, double* dwref, double* jac, int l, int col, const JacSpec& J, double deltaInv, double turbResScale,
                       hipStream_t s)
{
    hipLaunchKernelGGL(k_fd_extract, own_grid(b), dim3(JC_BX, JC_BY, 1), 0, s, b, dwref, jac, l, col, J, deltaInv, turbResScale);
}

void launch_fd_snap(const BlkView& b, double* snap, const JacSpec& J, double turbResScale, hipStream_t s)
{
    hipLaunchKernelGGL(k_fd_snap, own_grid(b), dim3(JC_BX, JC_BY, 1), 0, s, b, snap, J, turbResScale);
}
void launch_fd_scatter(const BlkView& b, const double* snap, double* jac, int l, const JacSpec& J, const double* dwref, double deltaInv,
                       hipStream_t s)
{
    const dim3 g = own_grid(b), t(JC_BX, JC_BY, 1);
    if (J.cn <= 7) hipLaunchKernelGGL(k_fd_scatter<7>, g, t, 0, s, b, snap, jac, l, J, dwref, deltaInv);
    else if (J.cn <= 13) hipLaunchKernelGGL(k_fd_scatter<13>, g, t, 0, s, b, snap, jac, l, J, dwref, deltaInv);
    else if (J.cn <= 27) hipLaunchKernelGGL(k_fd_scatter<27>, g, t, 0, s, b, snap, jac, l, J, dwref, deltaInv);
    else hipLaunchKernelGGL(k_fd_scatter<35>, g, t, 0, s, b, snap, jac, l, J, dwref, deltaInv);
}
