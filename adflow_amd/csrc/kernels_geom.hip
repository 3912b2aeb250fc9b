// Geometry pipeline on the device (SURVEY.md §8f "next" row 3): cell volumes, face
// normals and boundary normals from the node coordinates, for the mesh-warping
// (`useSpatial`) branch of blocketteRes (src/NKSolver/blockette.F90:203-211).
//
// Reference semantics (src/adjoint/adjointExtra.F90):
//   volume_block     :5-178    six pyramids per cell around the cell centre, |.|, then the
//                              halo-volume repair (a collapsed halo takes its neighbour's volume)
//   metric_block     :179-268  face normal = fact * (diagonal x diagonal), fact = +-1/2
//   boundaryNormals  :270-364  unit outward normal of every boundary-subface cell
// Pointwise gathers over nodes; roofline: HBM.
#include "internal.h"

#define GM_BX 64
#define GM_BY 4

struct P3 { double x, y, z; };

__device__ __forceinline__ P3 node(const BlkView& b, long n)
{
    P3 p;
    p.x = b.x[n]; p.y = b.x[n + b.nbox]; p.z = b.x[n + 2 * b.nbox];
    return p;
}

// volpym (adjointExtra.F90:157-176); (xp,yp,zp) = cell centre
__device__ __forceinline__ double volpym(const P3& c, const P3& a, const P3& b, const P3& cc, const P3& d)
{
    return (c.x - 0.25 * (a.x + b.x + cc.x + d.x)) * ((a.y - cc.y) * (b.z - d.z) - (a.z - cc.z) * (b.y - d.y)) +
           (c.y - 0.25 * (a.y + b.y + cc.y + d.y)) * ((a.z - cc.z) * (b.x - d.x) - (a.x - cc.x) * (b.z - d.z)) +
           (c.z - 0.25 * (a.z + b.z + cc.z + d.z)) * ((a.x - cc.x) * (b.y - d.y) - (a.y - cc.y) * (b.x - d.x));
}

// all box cells: vol = 0 outside 1..ie x 1..je x 1..ke
__global__ __launch_bounds__(GM_BX* GM_BY) void k_volume(BlkView b)
{
    const int i = blockIdx.x * GM_BX + threadIdx.x;
    const int j = blockIdx.y * GM_BY + threadIdx.y;
    const int k = blockIdx.z;
    if (i > b.ib || j > b.jb) return;
    const long c = b.idx(i, j, k);
    if (i < 1 || i > b.ie || j < 1 || j > b.je || k < 1 || k > b.ke) { b.vol[c] = 0.0; return; }
    const long si = 1, sj = b.ldi, sk = b.ldk;
    // nodes: (i|l, j|m, k|n) with l = i-1, m = j-1, n = k-1
    const P3 ijk = node(b, c), imk = node(b, c - sj), imn = node(b, c - sj - sk), ijn = node(b, c - sk);
    const P3 ljk = node(b, c - si), lmk = node(b, c - si - sj), lmn = node(b, c - si - sj - sk), ljn = node(b, c - si - sk);
    P3 ctr;
    ctr.x = 0.125 * (ijk.x + imk.x + imn.x + ijn.x + ljk.x + lmk.x + lmn.x + ljn.x);
    ctr.y = 0.125 * (ijk.y + imk.y + imn.y + ijn.y + ljk.y + lmk.y + lmn.y + ljn.y);
    ctr.z = 0.125 * (ijk.z + imk.z + imn.z + ijn.z + ljk.z + lmk.z + lmn.z + ljn.z);
    const double vp1 = volpym(ctr, ijk, ijn, imn, imk);
    const double vp2 = volpym(ctr, ljk, lmk, lmn, ljn);
    const double vp3 = volpym(ctr, ijk, ljk, ljn, ijn);
    const double vp4 = volpym(ctr, imk, imn, lmn, lmk);
    const double vp5 = volpym(ctr, ijk, imk, lmk, ljk);
    const double vp6 = volpym(ctr, ijn, ljn, lmn, imn);
    b.vol[c] = fabs((1.0 / 6.0) * (vp1 + vp2 + vp3 + vp4 + vp5 + vp6));
}

// halo-volume repair, one direction per launch (the later directions read what the earlier ones wrote)
template <int DIR>
__global__ __launch_bounds__(256) void k_volume_halo(BlkView b)
{
    const double haloCellRatio = 1e-10;
    const int a = blockIdx.x * 256 + threadIdx.x, bb = blockIdx.y;
    int lo, hi, lo2, hi2;
    long h1, n1, h2, n2;
    if (DIR == 0) {          // i faces: j = 2..jl, k = 2..kl
        const int j = a + 2, k = bb + 2;
        if (j > b.jl || k > b.kl) return;
        h1 = b.idx(1, j, k); n1 = b.idx(2, j, k); h2 = b.idx(b.ie, j, k); n2 = b.idx(b.il, j, k);
    } else if (DIR == 1) {   // j faces: i = 1..ie, k = 2..kl
        const int i = a + 1, k = bb + 2;
        if (i > b.ie || k > b.kl) return;
        h1 = b.idx(i, 1, k); n1 = b.idx(i, 2, k); h2 = b.idx(i, b.je, k); n2 = b.idx(i, b.jl, k);
    } else {                 // k faces: i = 1..ie, j = 1..je
        const int i = a + 1, j = bb + 1;
        if (i > b.ie || j > b.je) return;
        h1 = b.idx(i, j, 1); n1 = b.idx(i, j, 2); h2 = b.idx(i, j, b.ke); n2 = b.idx(i, j, b.kl);
    }
    (void)lo; (void)hi; (void)lo2; (void)hi2;
    if (b.vol[h1] / b.vol[n1] < haloCellRatio) b.vol[h1] = b.vol[n1];
    if (b.vol[h2] / b.vol[n2] < haloCellRatio) b.vol[h2] = b.vol[n2];
}

__device__ __forceinline__ void cross_store(double* __restrict__ s, long c, long nb, double fact, const P3& p1, const P3& p2,
                                            const P3& q1, const P3& q2)
{
    const double v1x = p1.x - p2.x, v1y = p1.y - p2.y, v1z = p1.z - p2.z;
    const double v2x = q1.x - q2.x, v2y = q1.y - q2.y, v2z = q1.z - q2.z;
    s[c] = fact * (v1y * v2z - v1z * v2y);
    s[c + nb] = fact * (v1z * v2x - v1x * v2z);
    s[c + 2 * nb] = fact * (v1x * v2y - v1y * v2x);
}

// sI (i = 0..ie, j = 1..je, k = 1..ke), sJ (1..ie, 0..je, 1..ke), sK (1..ie, 1..je, 0..ke)
__global__ __launch_bounds__(GM_BX* GM_BY) void k_metric(BlkView b, double fact)
{
    const int i = blockIdx.x * GM_BX + threadIdx.x;
    const int j = blockIdx.y * GM_BY + threadIdx.y;
    const int k = blockIdx.z;
    if (i > b.ie || j > b.je || k > b.ke) return;
    const long c = b.idx(i, j, k), nb = b.nbox;
    const long si = 1, sj = b.ldi, sk = b.ldk;
    if (j >= 1 && k >= 1)    // v1 = x(i,j,n) - x(i,m,k) ; v2 = x(i,j,k) - x(i,m,n)
        cross_store(b.sI, c, nb, fact, node(b, c - sk), node(b, c - sj), node(b, c), node(b, c - sj - sk));
    if (i >= 1 && k >= 1)    // v1 = x(i,j,n) - x(l,j,k) ; v2 = x(l,j,n) - x(i,j,k)
        cross_store(b.sJ, c, nb, fact, node(b, c - sk), node(b, c - si), node(b, c - si - sk), node(b, c));
    if (i >= 1 && j >= 1)    // v1 = x(i,j,k) - x(l,m,k) ; v2 = x(l,j,k) - x(i,m,k)
        cross_store(b.sK, c, nb, fact, node(b, c), node(b, c - si - sj), node(b, c - si), node(b, c - sj));
}

__global__ __launch_bounds__(256) void k_boundary_normals(BlkView b, BcFaceDev f, double* __restrict__ norm)
{
    const int isize = f.icEnd - f.icBeg + 1, jsize = f.jcEnd - f.jcBeg + 1;
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x, n = (long)isize * jsize;
    if (t >= n) return;
    const int i = f.icBeg + (int)(t % isize), j = f.jcBeg + (int)(t / isize);
    const double* s;
    long c;
    double mult;
    switch (f.faceID) {
    case ADFLOW_IMIN: mult = -1.0; s = b.sI; c = b.idx(1, i, j); break;
    case ADFLOW_IMAX: mult = 1.0; s = b.sI; c = b.idx(b.il, i, j); break;
    case ADFLOW_JMIN: mult = -1.0; s = b.sJ; c = b.idx(i, 1, j); break;
    case ADFLOW_JMAX: mult = 1.0; s = b.sJ; c = b.idx(i, b.jl, j); break;
    case ADFLOW_KMIN: mult = -1.0; s = b.sK; c = b.idx(i, j, 1); break;
    default: mult = 1.0; s = b.sK; c = b.idx(i, j, b.kl); break;
    }
    const double xxp = s[c], yyp = s[c + b.nbox], zzp = s[c + 2 * b.nbox];
    double fact = sqrt(xxp * xxp + yyp * yyp + zzp * zzp);
    if (fact > 0.0) fact = mult / fact;
    norm[t] = fact * xxp;
    norm[t + n] = fact * yyp;
    norm[t + 2 * n] = fact * zzp;
}

// ---------------------------------------------------------------------------
// xhalo_block (adjointExtra.F90:365-599): halo nodes by linear extrapolation, three ordered passes
//   0: i = 0 / ie   for j = 1..jl, k = 1..kl
//   1: j = 0 / je   for i = 0..ie, k = 1..kl   (reads what pass 0 wrote)
//   2: k = 0 / ke   for i = 0..ie, j = 0..je   (reads passes 0 and 1)
// then the mirror image in symmetry planes.  Level-batched: blockIdx.z = block slot.
// ---------------------------------------------------------------------------
template <int PASS>
__global__ __launch_bounds__(256) void k_xhalo(const BlkView* __restrict__ tab)
{
    const BlkView& b = tab[blockIdx.z + 1];
    if (b.nx == 0) return;
    const int a = blockIdx.x * 64 + threadIdx.x, c2 = blockIdx.y * 4 + threadIdx.y;
    const long nb = b.nbox;
    long h0, s0, h1, s1;     // halo node + step towards the interior, on the min and on the max side
    if (PASS == 0) {
        const int j = a + 1, k = c2 + 1;
        if (j > b.jl || k > b.kl) return;
        h0 = b.idx(0, j, k); s0 = 1; h1 = b.idx(b.ie, j, k); s1 = -1;
    } else if (PASS == 1) {
        const int i = a, k = c2 + 1;
        if (i > b.ie || k > b.kl) return;
        h0 = b.idx(i, 0, k); s0 = b.ldi; h1 = b.idx(i, b.je, k); s1 = -(long)b.ldi;
    } else {
        const int i = a, j = c2;
        if (i > b.ie || j > b.je) return;
        h0 = b.idx(i, j, 0); s0 = b.ldk; h1 = b.idx(i, j, b.ke); s1 = -(long)b.ldk;
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        double* x = b.x + m * nb;
        x[h0] = 2.0 * x[h0 + s0] - x[h0 + 2 * s0];
        x[h1] = 2.0 * x[h1 + s1] - x[h1 + 2 * s1];
    }
}

void launch_xhalo_level(const BlkView* tab, int nslots, int nx, int ny, int nz, hipStream_t s)
{
    LEVEL_SPLIT(nslots, nz + 4, launch_xhalo_level(tab + s0_, n_, nx, ny, nz, s));
    if (nslots <= 0) return;
    const int ie = nx + 2, je = ny + 2, jl = ny + 1, kl = nz + 1;
    dim3 blk(64, 4, 1);
    hipLaunchKernelGGL(k_xhalo<0>, dim3((jl + 63) / 64, (kl + 3) / 4, nslots), blk, 0, s, tab);
    hipLaunchKernelGGL(k_xhalo<1>, dim3((ie + 64) / 64, (kl + 3) / 4, nslots), blk, 0, s, tab);
    hipLaunchKernelGGL(k_xhalo<2>, dim3((ie + 64) / 64, (je + 4) / 4, nslots), blk, 0, s, tab);
}

// symmetry planes: halo node = mirror image of the second node plane, x(0) = x(2) + 2 ((x(1) - x(2)) . n) n, over the
// node range of the subface extended to 0 / max+1 where it reaches the edge of the block face (adjointExtra.F90:431-597)
__global__ __launch_bounds__(256) void k_xhalo_symm(const BlkView* __restrict__ tab, const BcEntry* __restrict__ ent,
                                                    const int* __restrict__ order)
{
    const BcEntry& e = ent[order[blockIdx.y]];
    const BcFaceDev& f = e.f;
    if (f.type != ADFLOW_BC_SYMM) return;
    const BlkView& b = tab[e.slot];
    double nx = f.symNorm[0], ny = f.symNorm[1], nz = f.symNorm[2];
    const double length = sqrt(nx * nx + ny * ny + nz * nz);
    nx /= length; ny /= length; nz /= length;
    if (!(length > 1.e-25)) return;          // eps of constants.F90: singular (collapsed) symmetry plane
    int r[4];
    bc_owned_range(f.faceID, f.icBeg, f.icEnd, f.jcBeg, f.jcEnd, b.il, b.jl, b.kl, r);
    const int iiMax = (f.faceID <= ADFLOW_IMAX) ? b.jl : b.il, jjMax = (f.faceID <= ADFLOW_JMAX) ? b.kl : b.jl;
    int iBeg = r[0] - 1, iEnd = r[1], jBeg = r[2] - 1, jEnd = r[3];      // node range of the subface
    if (iBeg == 1) iBeg = 0;
    if (iEnd == iiMax) iEnd = iiMax + 1;
    if (jBeg == 1) jBeg = 0;
    if (jEnd == jjMax) jEnd = jjMax + 1;
    const int na = iEnd - iBeg + 1, nbb = jEnd - jBeg + 1;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (na <= 0 || nbb <= 0 || t >= (long)na * nbb) return;
    const int a = iBeg + (int)(t % na), c2 = jBeg + (int)(t / na);
    long h, s;
    switch (f.faceID) {
    case ADFLOW_IMIN: h = b.idx(0, a, c2); s = 1; break;
    case ADFLOW_IMAX: h = b.idx(b.ie, a, c2); s = -1; break;
    case ADFLOW_JMIN: h = b.idx(a, 0, c2); s = b.ldi; break;
    case ADFLOW_JMAX: h = b.idx(a, b.je, c2); s = -(long)b.ldi; break;
    case ADFLOW_KMIN: h = b.idx(a, c2, 0); s = b.ldk; break;
    default: h = b.idx(a, c2, b.ke); s = -(long)b.ldk;
    }
    const long nb = b.nbox;
    const long n1 = h + s, n2 = h + 2 * s;
    const double v1 = b.x[n1] - b.x[n2], v2 = b.x[n1 + nb] - b.x[n2 + nb], v3 = b.x[n1 + 2 * nb] - b.x[n2 + 2 * nb];
    const double dot = 2.0 * (v1 * nx + v2 * ny + v3 * nz);
    b.x[h] = b.x[n2] + dot * nx;
    b.x[h + nb] = b.x[n2 + nb] + dot * ny;
    b.x[h + 2 * nb] = b.x[n2 + 2 * nb] + dot * nz;
}

void launch_xhalo_symm(const BlkView* tab, const BcEntry* ent, const int* order, const BcPhase& ph, hipStream_t s)
{
    if (ph.count <= 0) return;
    // the node range is at most (cells + 3) per direction: bound it by (sqrt(maxCells) + 3)^2 generously via maxCells * 4 + 64
    const long maxNodes = ph.maxCells * 4 + 64;
    hipLaunchKernelGGL(k_xhalo_symm, dim3((unsigned)((maxNodes + 255) / 256), ph.count, 1), dim3(256), 0, s, tab, ent, order + ph.first);
}

// coarseOwnedCoordinates (coarseUtils.F90:780-858): the nodes 1..il of a coarse block are the fine nodes kept by the
// coarsening.  With the transfer maps: coarse node m >= 2 closes coarse cell m, whose last fine cell is mgFine(m,2),
// so it is fine node mgFine(m,2); node 1 is fine node 1.
__global__ __launch_bounds__(256) void k_coarse_coordinates(const BlkView* __restrict__ ctab, const BlkView* __restrict__ ftab)
{
    const BlkView& c = ctab[blockIdx.z + 1];
    const BlkView& f = ftab[blockIdx.z + 1];
    if (c.nx == 0) return;
    const int i = blockIdx.x * 64 + threadIdx.x + 1;
    const int jk = blockIdx.y * 4 + threadIdx.y;
    const int j = jk % c.jl + 1, k = jk / c.jl + 1;
    if (i > c.il || k > c.kl) return;
    const int fi = (i == 1) ? 1 : c.mgIFine[2 * i + 1], fj = (j == 1) ? 1 : c.mgJFine[2 * j + 1],
              fk = (k == 1) ? 1 : c.mgKFine[2 * k + 1];
    const long cn = c.idx(i, j, k), fn = f.idx(fi, fj, fk);
#pragma unroll
    for (int m = 0; m < 3; ++m) c.x[cn + m * c.nbox] = f.x[fn + m * f.nbox];
}

void launch_coarse_coordinates_level(const BlkView* ctab, const BlkView* ftab, int nslots, int nx, int ny, int nz, hipStream_t s)
{
    LEVEL_SPLIT(nslots, nz + 4, launch_coarse_coordinates_level(ctab + s0_, ftab + s0_, n_, nx, ny, nz, s));
    if (nslots <= 0) return;
    const int il = nx + 1, jl = ny + 1, kl = nz + 1;
    hipLaunchKernelGGL(k_coarse_coordinates, dim3((il + 63) / 64, (jl * kl + 3) / 4, nslots), dim3(64, 4, 1), 0, s, ctab, ftab);
}

void launch_volume_metric(const BlkView& b, int rightHanded, hipStream_t s)
{
    const dim3 blk(GM_BX, GM_BY, 1);
    hipLaunchKernelGGL(k_volume, dim3((b.ib + GM_BX) / GM_BX, (b.jb + GM_BY) / GM_BY, b.kb + 1), blk, 0, s, b);
    hipLaunchKernelGGL((k_volume_halo<0>), dim3((b.ny + 255) / 256, b.nz), dim3(256), 0, s, b);
    hipLaunchKernelGGL((k_volume_halo<1>), dim3((b.ie + 255) / 256, b.nz), dim3(256), 0, s, b);
    hipLaunchKernelGGL((k_volume_halo<2>), dim3((b.ie + 255) / 256, b.je), dim3(256), 0, s, b);
    hipLaunchKernelGGL(k_metric, dim3((b.ie + GM_BX) / GM_BX, (b.je + GM_BY) / GM_BY, b.ke + 1), blk, 0, s, b,
                       rightHanded ? 0.5 : -0.5);
}

void launch_boundary_normals(const BlkView& b, const BcFaceDev* faces, int nBocos, hipStream_t s)
{
    for (int m = 0; m < nBocos; ++m) {
        if (!faces[m].norm) continue;
        const long n = (long)(faces[m].icEnd - faces[m].icBeg + 1) * (faces[m].jcEnd - faces[m].jcBeg + 1);
        hipLaunchKernelGGL(k_boundary_normals, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, b, faces[m],
                           const_cast<double*>(faces[m].norm));
    }
}

// ---- wallDistance::updateWallDistancesQuickly (src/wallDistance/wallDistance.F90:36-120) -------------------------------------
// d2Wall of the owned cells from the wall association found once on the host (determineWallAssociation, :1663-2002): the four
// surface nodes of the closest wall quad and the (u, v) of the closest point on it.  After a mesh warp only the surface
// coordinates xSurf (gathered by the host's VecScatter, updateXSurf :2004-2051) and the cell centre move.
// ind: (4, nx, ny, nz) 1-based node numbers, 0 in the first = no association (d2Wall = large); uv: (2, nx, ny, nz).
__global__ __launch_bounds__(GM_BX* GM_BY) void k_wall_distance(BlkView b, const int4* __restrict__ ind, const double2* __restrict__ uv,
                                                                const double* __restrict__ xSurf)
{
    const int i = blockIdx.x * GM_BX + threadIdx.x + 2;
    const int j = blockIdx.y * GM_BY + threadIdx.y + 2;
    const int k = blockIdx.z + 2;
    if (i > b.il || j > b.jl) return;
    const long c = b.idx(i, j, k);
    const long t = ((long)(k - 2) * b.ny + (j - 2)) * b.nx + (i - 2);
    const int4 n = ind[t];
    if (n.x == 0) { b.d2wall[c] = 1.e+37; return; }                        // constants::large
    const double u = uv[t].x, v = uv[t].y;
    const double w1 = (1.0 - u) * (1.0 - v), w2 = u * (1.0 - v), w3 = u * v, w4 = (1.0 - u) * v;
    const double* x1 = xSurf + 3l * (n.x - 1);
    const double* x2 = xSurf + 3l * (n.y - 1);
    const double* x3 = xSurf + 3l * (n.z - 1);
    const double* x4 = xSurf + 3l * (n.w - 1);
    const long si = 1, sj = b.ldi, sk = b.ldk;
    double d2 = 0.0;
    for (int q = 0; q < 3; ++q) {
        const double* xq = b.x + (long)q * b.nbox;
        const double xp = w1 * x1[q] + w2 * x2[q] + w3 * x3[q] + w4 * x4[q];
        const double xc = 0.125 * (xq[c - si - sj - sk] + xq[c - sj - sk] + xq[c - si - sk] + xq[c - sk] + xq[c - si - sj] + xq[c - sj] +
                                   xq[c - si] + xq[c]);
        d2 += (xc - xp) * (xc - xp);
    }
    b.d2wall[c] = sqrt(d2);
}

void launch_wall_distance(const BlkView& b, const int* ind, const double* uv, const double* xSurf, hipStream_t s)
{
    dim3 g((b.nx + GM_BX - 1) / GM_BX, (b.ny + GM_BY - 1) / GM_BY, b.nz);
    hipLaunchKernelGGL(k_wall_distance, g, dim3(GM_BX, GM_BY, 1), 0, s, b, (const int4*)ind, (const double2*)uv, xSurf);
}
