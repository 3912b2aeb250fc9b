// C-ABI entry points (include/adflow_gpu.h): block registry, HBM mirrors,
// host<->device transfers and the launch sequences of the hot path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "internal.h"

extern int g_march_kch, g_viscous_tiled, g_inviscid_march, g_roe_march, g_sa_march;
int g_test_fault = 0;        // tuning "test_fault" (tests only): bit 0 = the hipGraph capture of a multigrid cycle reports failure, bit 1 = the
                             // split evaluation fails behind its fork -- the error paths must leave the library usable
int g_rvec_joint = 1;        // tuning "rvec_joint": the six entries of a cell of the matrix-free residual vector written by ONE kernel (KParams::rvecTurbFromDw)
int g_pc_fused = 1;          // tuning "pc_fused": first-order Roe + thin-layer viscous flux of the preconditioner matrix as ONE march (kernels_pc_march.hip), plain and dual
int g_xcd_tiles = 2;        // tuning "xcd_tiles": 0 = tiles in launch order, 1 = XCD x owns the x-th eighth of the launch, 2 = of every round

namespace {

std::string g_err;
int g_device = -1;
hipStream_t g_stream = nullptr;
// side streams of blocketteRes: the SA residual and the nodal gradients are independent of the inviscid kernel (which is bound by
// FP64 issue while they are bound by HBM): launched on their own queues they share the CUs with it (tuning "overlap")
hipStream_t g_streamB = nullptr, g_streamC = nullptr;
hipEvent_t g_evFork = nullptr, g_evB = nullptr, g_evC = nullptr, g_evB1 = nullptr;
hipStream_t g_streamX = nullptr;                 // the RCCL send / recv group of a halo exchange
hipEvent_t g_evPack = nullptr, g_evComm = nullptr;
int g_overlap = 1;
adflow_opts g_opts;
bool g_have_opts = false;
hipEvent_t g_events[64];
bool g_events_ready = false;
bool g_async = false;   // adflow_gpu_set_async: entry points enqueue only

int fail(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct DevArr {
    double* base = nullptr;   // hipMalloc pointer
    int ncomp = 0;
};

struct Block {
    adflow_block_desc d;
    BlkView v;
    long boxsize = 0;         // ldi*(jb+1)*(kb+1)
    std::vector<void*> allocs;
    bool geom_uploaded = false;
    bool normals_from_x_ok = true;     // the uploaded sI / sJ / sK equal metric_block(x): the kernels may re-form them from the nodes
    unsigned geom_is_host = 0;         // bit 0..3: the device copy of x / sI / sJ / sK IS the descriptor's host array (set by an upload from
                                       // the registered pointer, cleared by an upload from any other buffer): normals_match_nodes reads the
                                       // HOST arrays, so it speaks for the device only when all four bits are set (round-4 advisor finding)
    bool face_vectors_valid = false;   // dI/dJ/dK derived from x
    bool ss_valid = false;    // entropy sensor variable matches the current state
    bool etot_consistent = false;   // owned-cell rhoE already equals computeEtotBlock(p, rho, v)
    std::vector<BcFaceDev> bc;      // boundary subfaces (device BCData), first nViscBocos = viscous walls
    std::vector<void*> bc_allocs;   // device copies of the BCData members: replaced at every bc_register
    int nViscBocos = 0;
    // coloured finite-difference Jacobian (adflow_gpu_fd_jacobian): reference state / residual, stencil blocks
    double *wref = nullptr, *dwref = nullptr, *jac = nullptr, *snap = nullptr;
    void *jac_raw = nullptr, *snap_raw = nullptr;
    int jac_ncomp = 0, snap_ncomp = 0;
    // wall association of updateWallDistancesQuickly: surfNodeIndices (4,nx,ny,nz), uv (2,nx,ny,nz)
    int* wd_ind = nullptr;
    double* wd_uv = nullptr;
    int wd_max_node = 0;               // largest 1-based surface node the association refers to
};

typedef std::tuple<int, int, int> Key;   // (level, sps, nn): iteration order = level, sps, nn
std::map<Key, Block*> g_blocks;

struct CommList {          // device index lists of one message / of the local copies
    int n = 0;
    int *blkA = nullptr, *blkB = nullptr;     // donor/halo block slots (pack/unpack use blkA only)
    long *offA = nullptr, *offB = nullptr;
    int peer = -1;
    double* buf = nullptr;                    // device message buffer (11 variables max)
};

struct CommPattern {
    bool present = false;
    CommList local;
    std::vector<CommList> sends, recvs;
    // host copies of the index triples until the block table is known
    std::vector<int32_t> h_donorBlock, h_donorIdx, h_haloBlock, h_haloIdx;
    std::vector<int32_t> h_sendProc, h_nsendCum, h_sendBlock, h_sendIdx;
    std::vector<int32_t> h_recvProc, h_nrecvCum, h_recvBlock, h_recvIdx;
    // periodic transformations (periodicDataType): halos they apply to
    struct Periodic {
        double rotMatrix[9], rotCenter[3], translation[3];
        std::vector<int32_t> h_block, h_idx;
        CommList list;
    };
    std::vector<Periodic> periodic;
    bool built = false;
};

struct ActRegion {       // one actuator region (actuatorRegionData.F90); the cell list addresses level-1 blocks
    std::vector<int32_t> h_block, h_idx;      // h_idx (n,3) column-major as make_list expects
    double force[3], heat, volume, relaxStart, relaxEnd;
    CommList list;
    bool built = false;
};
std::vector<ActRegion> g_act;

std::map<std::pair<int, int>, CommPattern> g_comm;   // (level, nLayers)
double* g_rvec_target = nullptr;     // residual vector the kernels of the evaluation in flight write (nk_residual_dev)
int g_rvec_done = 0;                  // bit 0: flow entries written, bit 1: turbulence entry written
int g_snap_done = 0;                  // the same for the snapshot entries of a coloured Jacobian evaluation (KParams::snapTab)
long g_state_gen = 0;        // bumped by every call that changes what a multigrid cycle enqueues (options, tuning, blocks, patterns, subfaces)
int g_split_eval = 1;       // tuning "split_eval": whalo2 + blocketteRes with the halo-free tiles inside the exchange: 1 = when the pattern has messages, 2 = always, 0 = never
int g_mg_graph = 1;         // tuning "mg_graph": a repeated adflow_gpu_mg_cycle is captured once into a hipGraph and replayed
int g_comm_self = 0;        // tuning "comm_self": same-process interfaces through pack / RCCL send+recv to self / unpack
int g_self_rank = 0;        // rank of this process in the RCCL communicator
std::map<int, BlkView*> g_tab;                        // level -> device table indexed by nn
std::map<int, int> g_tab_size;
std::map<int, std::pair<int4*, int>> g_tiles;          // level -> XCD-ordered tile table of the marching kernel
std::map<int, std::pair<int4*, int>> g_gf_tiles;       // level -> round-fitted chunk table of k_visc_gf
std::map<int, std::pair<int4*, int>> g_gf_tiles_int, g_gf_tiles_bnd;   // the same chunks: those that read no halo cell / the others
std::map<int, std::pair<int4*, int>> g_sa_tiles, g_sa_tiles_int, g_sa_tiles_bnd;   // the same three for k_sa_march
int g_num_cus = 0;
int g_gf_nofit = 0;         // tuning gf_cus = -1 (tests): chunks of march_kch planes instead of the round fit
int g_phase_base = 0;                                    // tuning "phase_events": first of 8 event slots, 0 = off
bool g_use_march = true;                               // tuning: adflow_gpu_set_tuning("euler_march", 0|1)
adflow_bc_callback g_bc_callback = nullptr;
adflow_bc_callback g_turb_bc_callback = nullptr;
double* g_norm_dev = nullptr;
int* g_floor_flag_dev = nullptr;      // raised by k_set_w_closures_level when a pressure hit its floor (FormFunction_mf)
int g_dadi_upd = 1;                  // tuning "dadi_upd": the D-ADI state update inside the k sweep's back substitution
int g_etot_flag_level = 0;            // > 0: the next whalo2 close on that level recomputes the owned energy only if the floor flag is up

void free_list(CommList& l)
{
    if (l.blkA) (void)hipFree(l.blkA);
    if (l.blkB) (void)hipFree(l.blkB);
    if (l.offA) (void)hipFree(l.offA);
    if (l.offB) (void)hipFree(l.offB);
    if (l.buf) (void)hipFree(l.buf);
    l = CommList();
}

// device lists of a pattern: rebuilt from the host copies at the next exchange
void drop_comm_lists(CommPattern& cp)
{
    free_list(cp.local);
    for (auto& l : cp.sends) free_list(l);
    for (auto& l : cp.recvs) free_list(l);
    for (auto& pd : cp.periodic) free_list(pd.list);
    cp.sends.clear();
    cp.recvs.clear();
    cp.built = false;
}

void invalidate_comm_level(int level)
{
    ++g_state_gen;
    if (level == 1)
        for (auto& r : g_act) r.built = false;      // cell offsets depend on the registered blocks
    for (auto& kv : g_comm)
        if (kv.first.first == level) drop_comm_lists(kv.second);
    auto it = g_tab.find(level);
    if (it != g_tab.end()) {
        (void)hipFree(it->second);
        g_tab.erase(it);
    }
    auto jt = g_tiles.find(level);
    if (jt != g_tiles.end()) {
        (void)hipFree(jt->second.first);
        g_tiles.erase(jt);
    }
    for (auto* mp : {&g_gf_tiles, &g_gf_tiles_int, &g_gf_tiles_bnd, &g_sa_tiles, &g_sa_tiles_int, &g_sa_tiles_bnd}) {
        jt = mp->find(level);
        if (jt != mp->end()) {
            (void)hipFree(jt->second.first);
            mp->erase(jt);
        }
    }
}


int ensure_table(int level);
int res_averaging_level(int level, const KParams& kp, double scaleDtl = 0.0);
int time_step_level(int level, const KParams& kp);
struct LevelTab { const BlkView* tab; int n, nx, ny, nz; };
int level_tab(int level, LevelTab* t);
int ensure_tiles(int level);
int ensure_gf_tiles(int level);
int ensure_sa_tiles(int level);
int build_comm(int level, int nLayers, CommPattern** out);
int halo_mask(int varStart, int varEnd, int commPressure, int commVisc, unsigned* mask, int* nvar);
int make_list(int level, const int32_t* blk, const int32_t* idx, int ld, int first, int n, int** d_blk, long** d_off);

Block* find_block(int nn, int level, int sps)
{
    auto it = g_blocks.find(Key(level, sps, nn));
    return it == g_blocks.end() ? nullptr : it->second;
}

int alloc_arr(Block* b, double** p, int ncomp)
{
    size_t bytes = (size_t)b->v.nbox * ncomp * sizeof(double) + 256;
    void* raw = nullptr;
    HIPCHK(hipMalloc(&raw, bytes));
    HIPCHK(hipMemsetAsync(raw, 0, bytes, g_stream));
    b->allocs.push_back(raw);
    *p = (double*)raw + ADF_PAD0;
    return 0;
}

// pinned staging buffer (grown on demand) for host <-> device box transfers
double* g_stage = nullptr;
size_t g_stage_elems = 0;

int stage_reserve(size_t elems)
{
    if (elems <= g_stage_elems) return 0;
    if (g_stage) (void)hipHostFree(g_stage);
    g_stage = nullptr;
    g_stage_elems = 0;
    HIPCHK(hipHostMalloc((void**)&g_stage, elems * sizeof(double), hipHostMallocDefault));
    g_stage_elems = elems;
    return 0;
}

// host sub-box (lo..lo+n-1 in each direction, column-major, ncomp components)
// <-> device box component(s).  The host side is the reference's pageable
// Fortran array; rows are (un)packed through one pinned buffer holding the
// k-slab range of the padded device box, moved with a single DMA per component.
int copy_box(Block* b, double* dev, const double* host_c, int ncomp, int lo_i, int n_i, int lo_j, int n_j, int lo_k, int n_k,
             bool to_device)
{
    if (!host_c) return 0;
    double* host = const_cast<double*>(host_c);
    const BlkView& v = b->v;
    const size_t slab = (size_t)v.ldk * n_k;            // padded elements covering k = lo_k .. lo_k+n_k-1
    if (stage_reserve(slab)) return 1;
    for (int c = 0; c < ncomp; ++c) {
        double* dptr = dev + (size_t)c * v.nbox + v.idx(0, 0, lo_k);
        double* hcomp = host + (size_t)c * n_i * n_j * n_k;
        if (to_device) {
            // read-modify-write of the slab keeps device values outside the sub-box
            const bool partial = (lo_i != 0 || lo_j != 0 || n_i != v.ib + 1 || n_j != v.jb + 1);
            if (partial) {
                HIPCHK(hipMemcpyAsync(g_stage, dptr, slab * 8, hipMemcpyDeviceToHost, g_stream));
                HIPCHK(hipStreamSynchronize(g_stream));
            }
            for (int k = 0; k < n_k; ++k)
                for (int j = 0; j < n_j; ++j)
                    memcpy(g_stage + (size_t)k * v.ldk + (size_t)(j + lo_j) * v.ldi + lo_i,
                           hcomp + ((size_t)k * n_j + j) * n_i, (size_t)n_i * 8);
            HIPCHK(hipMemcpyAsync(dptr, g_stage, slab * 8, hipMemcpyHostToDevice, g_stream));
            HIPCHK(hipStreamSynchronize(g_stream));
        } else {
            HIPCHK(hipMemcpyAsync(g_stage, dptr, slab * 8, hipMemcpyDeviceToHost, g_stream));
            HIPCHK(hipStreamSynchronize(g_stream));
            for (int k = 0; k < n_k; ++k)
                for (int j = 0; j < n_j; ++j)
                    memcpy(hcomp + ((size_t)k * n_j + j) * n_i,
                           g_stage + (size_t)k * v.ldk + (size_t)(j + lo_j) * v.ldi + lo_i, (size_t)n_i * 8);
        }
    }
    return 0;
}

int g_lumped = 0;       // inputDiscretization::lumpedDiss while a preconditioner matrix is assembled
int g_metric_from_x = 5;   // tuning "metric_from_x": bit 0 the SA march, bit 2 the time-step kernel re-form the face normals from the node coordinates.
                           // (k_visc_gf and k_roe_march keep the stored normals: the node form costs more registers than either has --
                           // profiles/r06_a_xn_2wg_*, r02_ba_variants.txt)

// the snapshot request of a Jacobian assembly (KParams::snapTab): set around the coloured evaluations of adflow_gpu_fd_jacobian
struct SnapReq { bool on = false; SnapSlot* dev = nullptr; int devSlots = 0; int level = 0, col = 0, l0 = 0, n = 0; double turbScale = 1.0; };
static SnapReq g_snapreq;
int g_jac_snap = 1;         // tuning "jac_snap": the marching kernels of the preconditioner matrix write the snapshots themselves (0: k_fd_snap / k_ad_snap)

KParams make_kparams(int level, double rFil, int fwMode)
{
    const adflow_opts& o = g_opts;
    KParams k;
    memset(&k, 0, sizeof k);
    k.equations = o.equations;
    k.spaceDiscr = (level == 1) ? o.spaceDiscr : o.spaceDiscrCoarse;
    k.limiter = o.limiter;
    k.orderTurb = o.orderTurb;
    k.turbProd = o.turbProd;
    k.viscous = (o.equations == ADFLOW_NS || o.equations == ADFLOW_RANS);
    k.eddyModel = (o.equations == ADFLOW_RANS);
    k.dirScaling = o.dirScaling;
    k.lumpedDiss = g_lumped;
    k.metricFromX = g_metric_from_x;
    if (k.metricFromX)
        for (auto& kv : g_blocks)
            if (std::get<0>(kv.first) == level && !kv.second->normals_from_x_ok) { k.metricFromX = 0; break; }
    k.sigma = o.sigma;
    k.useQCR = o.useQCR;
    k.useRotationSA = o.useRotationSA;
    k.useft2SA = o.useft2SA;
    k.fineGrid = (level == o.groundLevel);
    k.groundLevelIsOne = (o.groundLevel == 1);
    k.doScaling = (o.dirScaling && level <= o.groundLevel);
    k.coarseInit = (level != o.groundLevel);
    k.fwMode = fwMode;
    k.updateEddy = (level <= o.groundLevel);
    k.rFil = rFil;
    k.sfil = 1.0 - rFil;
    k.vis2 = o.vis2; k.vis4 = o.vis4; k.vis2Coarse = o.vis2Coarse; k.adis = o.adis;
    k.acousticScaleFactor = o.acousticScaleFactor; k.kappaCoef = o.kappaCoef;
    k.gammaConstant = o.gammaConstant; k.gammaInf = o.gammaInf; k.pInf = o.pInf; k.pInfCorr = o.pInfCorr; k.rhoInf = o.rhoInf; k.RGas = o.RGas;
    k.muRef = o.muRef; k.TRef = o.TRef; k.timeRef = o.timeRef;
    k.prandtl = o.prandtl; k.prandtlTurb = o.prandtlTurb;
    k.SSuthDim = o.SSuthDim; k.muSuthDim = o.muSuthDim; k.TSuthDim = o.TSuthDim;
    k.sa_k = o.SAKappa; k.sa_cb1 = o.SAcb1; k.sa_cb2 = o.SAcb2; k.sa_cb3 = o.SAsigma; k.sa_cv1 = o.SAcv1;
    k.sa_cw1 = o.SAcw1; k.sa_cw2 = o.SAcw2; k.sa_cw3 = o.SAcw3; k.sa_ct3 = o.SAct3; k.sa_ct4 = o.SAct4;
    k.sa_crot = o.SAcrot;
    k.sa_qqFactor = (o.turbRelax == 2) ? 1.0 + (1.0 - o.alfaTurb) / o.alfaTurb : 1.0;
    k.sa_updFactor = (o.turbRelax == 1) ? o.alfaTurb : 1.0;
    k.cfl = (level == 1) ? o.cfl : o.cflCoarse;
    k.cflLimit = o.cflLimit; k.smoop = o.smoop; k.fcoll = o.fcoll; k.turbResScale = o.turbResScale;
    for (int i = 0; i < 10; ++i) k.wInf[i] = o.wInf[i];
    if (g_snapreq.on && level == g_snapreq.level) {
        k.snapTab = g_snapreq.dev; k.snapCol = g_snapreq.col; k.snapL0 = g_snapreq.l0; k.snapN = g_snapreq.n;
        k.snapTurbScale = g_snapreq.turbScale;
    }
    return k;
}

int need_ready(void)
{
    if (g_device < 0) return fail("adflow_gpu_init has not been called");
    if (!g_have_opts) return fail("adflow_gpu_set_options has not been called");
    return 0;
}

template <typename Fn>
int for_level(int level, Fn fn)
{
    bool any = false;
    for (auto& kv : g_blocks) {
        if (std::get<0>(kv.first) != level) continue;
        any = true;
        int rc = fn(kv.second);
        if (rc) return rc;
    }
    if (!any) return fail("no block registered on level %d", level);
    return 0;
}

}  // namespace
// instrumentation of blocketteRes (tuning "phase_events" = first event slot): marks 0 entry, 1 closures / BCs / halos done,
// 2 time step, 3 SA residual, 4 inviscid fluxes, 5 nodal gradients, 6 viscous fluxes + sources (end); with the viscous march in
// front of the Roe march (tuning visc_first): 4 nodal gradients, 5 viscous fluxes, 6 inviscid fluxes + sources
void adf_phase_mark(int i)
{
    static unsigned hit = 0;
    if (g_phase_base <= 0 || !g_stream) return;
    if (i == 0) hit = 0;
    if (i == 6)      // phases this configuration does not have (no viscous part, fused kernels): zero-length, closed here
        for (int m = 1; m < 6; ++m)
            if (!(hit & (1u << m))) (void)hipEventRecord(g_events[g_phase_base + m], g_stream);
    hit |= 1u << i;
    (void)hipEventRecord(g_events[g_phase_base + i], g_stream);
}
namespace {
inline void phase_mark(int i) { adf_phase_mark(i); }

int sync_and_check(void)
{
    HIPCHK(hipGetLastError());
    if (!g_async) HIPCHK(hipStreamSynchronize(g_stream));
    return 0;
}

}  // namespace

extern "C" {

const char* adflow_gpu_last_error(void) { return g_err.c_str(); }

int adflow_gpu_init(int device_ordinal)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return fail("no HIP device visible: the MI355X engine has no CPU fallback");
    if (device_ordinal < 0 || device_ordinal >= n) return fail("device ordinal %d out of range (%d devices)", device_ordinal, n);
    HIPCHK(hipSetDevice(device_ordinal));
    if (!g_stream) HIPCHK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    if (!g_streamB) {
        HIPCHK(hipStreamCreateWithFlags(&g_streamB, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&g_streamC, hipStreamNonBlocking));
        HIPCHK(hipEventCreate(&g_evFork)); HIPCHK(hipEventCreate(&g_evB)); HIPCHK(hipEventCreate(&g_evC)); HIPCHK(hipEventCreate(&g_evB1));
        HIPCHK(hipStreamCreateWithFlags(&g_streamX, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&g_evPack, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&g_evComm, hipEventDisableTiming));
    }
    if (!g_events_ready) {
        for (int i = 0; i < 64; ++i) HIPCHK(hipEventCreate(&g_events[i]));
        g_events_ready = true;
    }
    g_device = device_ordinal;
    return 0;
}

int adflow_gpu_device_name(char* buf, int len)
{
    if (g_device < 0) return fail("adflow_gpu_init has not been called");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, g_device));
    snprintf(buf, len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return 0;
}

static void free_side_buffers(void);

int adflow_gpu_finalize(void)
{
    // device tables, tile tables, communication patterns, boundary plans and every block: the same path as release_all, so
    // that an init after a finalize starts from an empty registry
    if (g_device >= 0) (void)adflow_gpu_release_all();
    free_side_buffers();
    if (g_events_ready) {
        for (int i = 0; i < 64; ++i) (void)hipEventDestroy(g_events[i]);
        g_events_ready = false;
    }
    if (g_stage) {
        (void)hipHostFree(g_stage);
        g_stage = nullptr;
        g_stage_elems = 0;
    }
    if (g_streamB) {
        (void)hipStreamDestroy(g_streamB); (void)hipStreamDestroy(g_streamC); (void)hipStreamDestroy(g_streamX);
        (void)hipEventDestroy(g_evFork); (void)hipEventDestroy(g_evB); (void)hipEventDestroy(g_evC); (void)hipEventDestroy(g_evB1);
        (void)hipEventDestroy(g_evPack); (void)hipEventDestroy(g_evComm);
        g_streamB = g_streamC = g_streamX = nullptr;
    }
    if (g_stream) {
        (void)hipStreamDestroy(g_stream);
        g_stream = nullptr;
    }
    g_device = -1;
    g_have_opts = false;
    return 0;
}

int adflow_gpu_set_options(const adflow_opts* o)
{
    if (!o) return fail("null options");
    if (o->equations < ADFLOW_EULER || o->equations > ADFLOW_RANS) return fail("equations=%d not supported", o->equations);
    if (o->nRKStages < 1 || o->nRKStages > ADFLOW_MAX_RK_STAGES) return fail("nRKStages=%d out of range", o->nRKStages);
    if (o->unsupported)
        return fail("configuration outside the GPU path:%s%s%s%s (this library computes the steady, constant-cp, 1-to-1 multiblock residual only)",
                    (o->unsupported & 1) ? " unsteady / time-spectral equationMode;" : "", (o->unsupported & 2) ? " cpModel /= cpConstant;" : "",
                    (o->unsupported & 4) ? " wall functions;" : "", (o->unsupported & 8) ? " overset blocks;" : "");
    if (o->equations == ADFLOW_RANS && o->turbModel != 2)   // spalartAllmaras (constants.F90)
        return fail("turbModel=%d not supported (Spalart-Allmaras only)", o->turbModel);
    if (memcmp(&g_opts, o, sizeof g_opts) != 0) ++g_state_gen;
    g_opts = *o;
    g_have_opts = true;
    return 0;
}

int adflow_gpu_block_register(int nn, int level, int sps, const adflow_block_desc* d)
{
    if (g_device < 0) return fail("adflow_gpu_init has not been called");
    if (!d) return fail("null block descriptor");
    if (d->nx < 1 || d->ny < 1 || d->nz < 1) return fail("block %d: bad dimensions %d %d %d", nn, d->nx, d->ny, d->nz);
    if (d->nw != 5 && d->nw != 6) return fail("block %d: nw=%d not supported (5, or 6 with SA)", nn, d->nw);
    if (find_block(nn, level, sps)) return fail("block (%d,%d,%d) already registered", nn, level, sps);
    // the level-batched kernels hold one spectral instance: a second one would silently keep stale dw / w
    if (sps != 1) return fail("block (%d,%d,%d): only sps = 1 (steady, one spectral instance) is supported", nn, level, sps);
    Block* b = new Block;
    b->d = *d;
    BlkView& v = b->v;
    memset(&v, 0, sizeof v);
    v.nx = d->nx; v.ny = d->ny; v.nz = d->nz; v.nw = d->nw;
    v.il = v.nx + 1; v.jl = v.ny + 1; v.kl = v.nz + 1;
    v.ie = v.nx + 2; v.je = v.ny + 2; v.ke = v.nz + 2;
    v.ib = v.nx + 3; v.jb = v.ny + 3; v.kb = v.nz + 3;
    v.ldi = ((v.ib + 1 + 15) / 16) * 16;
    v.ldk = v.ldi * (v.jb + 1);
    v.mfact = d->rightHanded ? 0.5 : -0.5;
    b->boxsize = (long)v.ldk * (v.kb + 1);
    v.nbox = ((b->boxsize + 15) / 16) * 16 + 16;
    int rc = 0;
    rc |= alloc_arr(b, &v.w, v.nw);
    rc |= alloc_arr(b, &v.p, 1);
    rc |= alloc_arr(b, &v.gamma, 1);
    rc |= alloc_arr(b, &v.rlv, 1);
    rc |= alloc_arr(b, &v.rev, 1);
    rc |= alloc_arr(b, &v.x, 3);
    rc |= alloc_arr(b, &v.sI, 3);
    rc |= alloc_arr(b, &v.sJ, 3);
    rc |= alloc_arr(b, &v.sK, 3);
    rc |= alloc_arr(b, &v.vol, 1);
    rc |= alloc_arr(b, &v.volRef, 1);
    rc |= alloc_arr(b, &v.d2wall, 1);
    if (d->addGridVelocities) {
        if (!(d->sFaceI && d->sFaceJ && d->sFaceK)) return fail("block (%d,%d,%d): addGridVelocities without sFaceI/J/K", nn, level, sps);
        rc |= alloc_arr(b, &v.sFace, 3);
    }
    v.moving = d->blockIsMoving ? 1 : 0;
    for (int m = 0; m < 3; ++m) v.rot[m] = d->rotRate[m];
    rc |= alloc_arr(b, &v.dI, 3);
    rc |= alloc_arr(b, &v.dJ, 3);
    rc |= alloc_arr(b, &v.dK, 3);
    rc |= alloc_arr(b, &v.xc, 3);
    rc |= alloc_arr(b, &v.dw, v.nw);
    rc |= alloc_arr(b, &v.fw, 5);
    rc |= alloc_arr(b, &v.dtl, 1);
    rc |= alloc_arr(b, &v.radI, 1);
    rc |= alloc_arr(b, &v.radJ, 1);
    rc |= alloc_arr(b, &v.radK, 1);
    rc |= alloc_arr(b, &v.ss, 1);
    rc |= alloc_arr(b, &v.aa, 1);
    rc |= alloc_arr(b, &v.grad, 12);
    rc |= alloc_arr(b, &v.scratch, 10);
    rc |= alloc_arr(b, &v.wn, 5);
    rc |= alloc_arr(b, &v.pn, 1);
    rc |= alloc_arr(b, &v.w1, 5);
    rc |= alloc_arr(b, &v.p1, 1);
    rc |= alloc_arr(b, &v.wr, 5);
    if (rc) return rc;
    {
        void* raw = nullptr;
        HIPCHK(hipMalloc(&raw, (size_t)v.nbox + 256));
        HIPCHK(hipMemsetAsync(raw, 0, (size_t)v.nbox + 256, g_stream));
        b->allocs.push_back(raw);
        v.flags = (uint8_t*)raw + 16;
    }
    // multigrid maps: small index arrays, re-laid out as [2*m + q]
    {
        auto up_pair = [&](const int32_t* h, int lo, int n, int** dev) -> int {
            *dev = nullptr;
            if (!h) return 0;
            std::vector<int> t((size_t)2 * (lo + n + 1), 0);
            for (int m = 0; m < n; ++m)
                for (int q = 0; q < 2; ++q) t[(size_t)2 * (lo + m) + q] = h[(size_t)m + (size_t)n * q];
            void* raw = nullptr;
            HIPCHK(hipMalloc(&raw, sizeof(int) * t.size()));
            HIPCHK(hipMemcpy(raw, t.data(), sizeof(int) * t.size(), hipMemcpyHostToDevice));
            b->allocs.push_back(raw);
            *dev = (int*)raw;
            return 0;
        };
        auto up_w = [&](const double* h, int lo, int n, double** dev) -> int {
            *dev = nullptr;
            if (!h) return 0;
            std::vector<double> t((size_t)(lo + n + 1), 0.0);
            for (int m = 0; m < n; ++m) t[(size_t)lo + m] = h[m];
            void* raw = nullptr;
            HIPCHK(hipMalloc(&raw, sizeof(double) * t.size()));
            HIPCHK(hipMemcpy(raw, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
            b->allocs.push_back(raw);
            *dev = (double*)raw;
            return 0;
        };
        if (up_pair(d->mgIFine, 1, v.ie, &v.mgIFine) || up_pair(d->mgJFine, 1, v.je, &v.mgJFine) ||
            up_pair(d->mgKFine, 1, v.ke, &v.mgKFine) || up_w(d->mgIWeight, 2, v.nx, &v.mgIWeight) ||
            up_w(d->mgJWeight, 2, v.ny, &v.mgJWeight) || up_w(d->mgKWeight, 2, v.nz, &v.mgKWeight) ||
            up_pair(d->mgICoarse, 2, v.nx, &v.mgICoarse) || up_pair(d->mgJCoarse, 2, v.ny, &v.mgJCoarse) ||
            up_pair(d->mgKCoarse, 2, v.nz, &v.mgKCoarse))
            return 1;
    }
    HIPCHK(hipStreamSynchronize(g_stream));
    g_blocks[Key(level, sps, nn)] = b;
    invalidate_comm_level(level);
    return 0;
}

static void bc_plan_drop(int level);
static void bc_plan_drop_all();
static void ad_cache_drop();      // the cached dual arrays of the forward-mode assembly refer to the blocks

int adflow_gpu_block_release(int nn, int level, int sps)
{
    auto it = g_blocks.find(Key(level, sps, nn));
    if (it == g_blocks.end()) return fail("block (%d,%d,%d) not registered", nn, level, sps);
    if (g_stream) (void)hipStreamSynchronize(g_stream);
    bc_plan_drop(level);
    ad_cache_drop();
    for (void* p : it->second->allocs) (void)hipFree(p);
    if (it->second->jac_raw) (void)hipFree(it->second->jac_raw);
    if (it->second->snap_raw) (void)hipFree(it->second->snap_raw);
    for (void* p : it->second->bc_allocs) (void)hipFree(p);
    delete it->second;
    g_blocks.erase(it);
    invalidate_comm_level(level);
    return 0;
}

int adflow_gpu_release_all(void)
{
    if (g_stream) (void)hipStreamSynchronize(g_stream);
    bc_plan_drop_all();
    ad_cache_drop();
    for (auto& kv : g_blocks) {
        for (void* p : kv.second->allocs) (void)hipFree(p);
        if (kv.second->jac_raw) (void)hipFree(kv.second->jac_raw);
        if (kv.second->snap_raw) (void)hipFree(kv.second->snap_raw);
        for (void* p : kv.second->bc_allocs) (void)hipFree(p);
        delete kv.second;
    }
    g_blocks.clear();
    {
        std::vector<int> levels;
        for (auto& kv : g_tab) levels.push_back(kv.first);
        for (int l : levels) invalidate_comm_level(l);
        g_comm.clear();
    }
    return 0;
}

// The marching kernels may re-form face normals from the node coordinates (tuning metric_from_x).  That is only the same
// computation if the host's sI / sJ / sK ARE metric_block(x) (adjointExtra.F90:176-268), which holds for every mesh the reference
// builds; a sample of faces of each array is checked against the cross products at upload, and a block that fails keeps the
// level on the stored normals instead of silently evaluating a different geometry.
static bool normals_match_nodes(const Block* b)
{
    const adflow_block_desc& d = b->d;
    const BlkView& v = b->v;
    if (!d.x || !d.sI || !d.sJ || !d.sK) return false;
    const double fact = d.rightHanded ? 0.5 : -0.5;
    const size_t nxn = (size_t)v.ie + 1, nyn = (size_t)v.je + 1, nzn = (size_t)v.ke + 1;        // x(0:ie, 0:je, 0:ke, 3)
    auto X = [&](int i, int j, int k, int q) { return d.x[(size_t)i + nxn * ((size_t)j + nyn * ((size_t)k + nzn * q))]; };
    auto cross = [&](const double p1[3], const double p2[3], const double q1[3], const double q2[3], double s[3]) {
        const double a[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]}, c[3] = {q1[0] - q2[0], q1[1] - q2[1], q1[2] - q2[2]};
        s[0] = fact * (a[1] * c[2] - a[2] * c[1]); s[1] = fact * (a[2] * c[0] - a[0] * c[2]); s[2] = fact * (a[0] * c[1] - a[1] * c[0]);
    };
    auto node = [&](int i, int j, int k, double p[3]) { for (int q = 0; q < 3; ++q) p[q] = X(i, j, k, q); };
    double worst = 0.0, scale = 0.0;
    const int ns = 5;
    for (int a = 0; a < ns; ++a)
        for (int c = 0; c < ns; ++c)
            for (int e = 0; e < ns; ++e) {
                // sample points spread over the index ranges incl. the first and last faces
                const int i = 1 + (int)((long)(v.ie - 1) * a / (ns - 1)), j = 1 + (int)((long)(v.je - 1) * c / (ns - 1)),
                          k = 1 + (int)((long)(v.ke - 1) * e / (ns - 1));
                double n11[3], n10[3], n01[3], n00[3], s[3];
                // sI(i, j, k), i = 0..ie, j = 1..je, k = 1..ke: nodes (i, j-1..j, k-1..k)
                node(i, j, k - 1, n11); node(i, j - 1, k, n10); node(i, j, k, n01); node(i, j - 1, k - 1, n00);
                cross(n11, n10, n01, n00, s);
                for (int q = 0; q < 3; ++q) {
                    const double h = d.sI[(size_t)i + nxn * ((size_t)(j - 1) + (size_t)v.je * ((size_t)(k - 1) + (size_t)v.ke * q))];
                    worst = std::max(worst, fabs(h - s[q])); scale = std::max(scale, fabs(h));
                }
                // sJ(i, j, k), i = 1..ie, j = 0..je, k = 1..ke: v1 = x(i,j,n) - x(l,j,k) ; v2 = x(l,j,n) - x(i,j,k)
                node(i, j, k - 1, n11); node(i - 1, j, k, n10); node(i - 1, j, k - 1, n01); node(i, j, k, n00);
                cross(n11, n10, n01, n00, s);
                for (int q = 0; q < 3; ++q) {
                    const double h = d.sJ[(size_t)(i - 1) + (size_t)v.ie * ((size_t)j + nyn * ((size_t)(k - 1) + (size_t)v.ke * q))];
                    worst = std::max(worst, fabs(h - s[q])); scale = std::max(scale, fabs(h));
                }
                // sK(i, j, k), i = 1..ie, j = 1..je, k = 0..ke: v1 = x(i,j,k) - x(l,m,k) ; v2 = x(l,j,k) - x(i,m,k)
                node(i, j, k, n11); node(i - 1, j - 1, k, n10); node(i - 1, j, k, n01); node(i, j - 1, k, n00);
                cross(n11, n10, n01, n00, s);
                for (int q = 0; q < 3; ++q) {
                    const double h = d.sK[(size_t)(i - 1) + (size_t)v.ie * ((size_t)(j - 1) + (size_t)v.je * ((size_t)k + nzn * q))];
                    worst = std::max(worst, fabs(h - s[q])); scale = std::max(scale, fabs(h));
                }
            }
    return worst <= 1.e-11 * std::max(scale, 1.e-300);
}

int adflow_gpu_upload_geometry(int nn, int level, int sps)
{
    Block* b = find_block(nn, level, sps);
    if (!b) return fail("block (%d,%d,%d) not registered", nn, level, sps);
    const adflow_block_desc& d = b->d;
    BlkView& v = b->v;
    // the marching kernels re-form face normals from the node coordinates and the viscous kernel derives its face vectors from
    // them: x is as much part of the geometry as the normals and the volumes
    if (!d.x || !d.sI || !d.sJ || !d.sK || !d.vol)
        return fail("block (%d,%d,%d): x, sI, sJ, sK and vol host arrays are required", nn, level, sps);
    int rc = 0;
    rc |= copy_box(b, v.x, d.x, 3, 0, v.ie + 1, 0, v.je + 1, 0, v.ke + 1, true);
    rc |= copy_box(b, v.sI, d.sI, 3, 0, v.ie + 1, 1, v.je, 1, v.ke, true);
    rc |= copy_box(b, v.sJ, d.sJ, 3, 1, v.ie, 0, v.je + 1, 1, v.ke, true);
    rc |= copy_box(b, v.sK, d.sK, 3, 1, v.ie, 1, v.je, 0, v.ke + 1, true);
    if (v.sFace) {
        rc |= copy_box(b, v.sFace, d.sFaceI, 1, 0, v.ie + 1, 1, v.je, 1, v.ke, true);
        rc |= copy_box(b, v.sFace + v.nbox, d.sFaceJ, 1, 1, v.ie, 0, v.je + 1, 1, v.ke, true);
        rc |= copy_box(b, v.sFace + 2 * v.nbox, d.sFaceK, 1, 1, v.ie, 1, v.je, 0, v.ke + 1, true);
    }
    rc |= copy_box(b, v.vol, d.vol, 1, 0, v.ib + 1, 0, v.jb + 1, 0, v.kb + 1, true);
    rc |= copy_box(b, v.volRef, d.volRef ? d.volRef : d.vol, 1, 0, v.ib + 1, 0, v.jb + 1, 0, v.kb + 1, true);
    rc |= copy_box(b, v.d2wall, d.d2Wall, 1, 2, v.nx, 2, v.ny, 2, v.nz, true);
    if (rc) return rc;
    // pack porosities + iblank into one byte per cell (internal.h)
    std::vector<uint8_t> f((size_t)v.nbox, 0);
    auto at = [&](int i, int j, int k) -> uint8_t& { return f[(size_t)v.idx(i, j, k)]; };
    if (d.porI)
        for (int k = 2; k <= v.kl; ++k)
            for (int j = 2; j <= v.jl; ++j)
                for (int i = 1; i <= v.il; ++i)
                    at(i, j, k) |= (uint8_t)((d.porI[(size_t)(i - 1) + (size_t)v.il * ((j - 2) + (size_t)v.ny * (k - 2))] + 1) & 3);
    if (d.porJ)
        for (int k = 2; k <= v.kl; ++k)
            for (int j = 1; j <= v.jl; ++j)
                for (int i = 2; i <= v.il; ++i)
                    at(i, j, k) |= (uint8_t)(((d.porJ[(size_t)(i - 2) + (size_t)v.nx * ((j - 1) + (size_t)v.jl * (k - 2))] + 1) & 3) << 2);
    if (d.porK)
        for (int k = 1; k <= v.kl; ++k)
            for (int j = 2; j <= v.jl; ++j)
                for (int i = 2; i <= v.il; ++i)
                    at(i, j, k) |= (uint8_t)(((d.porK[(size_t)(i - 2) + (size_t)v.nx * ((j - 2) + (size_t)v.ny * (k - 1))] + 1) & 3) << 4);
    for (int k = 0; k <= v.kb; ++k)
        for (int j = 0; j <= v.jb; ++j)
            for (int i = 0; i <= v.ib; ++i) {
                int ibl = d.iblank ? d.iblank[(size_t)i + (size_t)(v.ib + 1) * (j + (size_t)(v.jb + 1) * k)] : 1;
                if (ibl > 0) at(i, j, k) |= 64;
            }
    HIPCHK(hipMemcpyAsync(v.flags, f.data(), (size_t)b->boxsize, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    b->geom_uploaded = true;
    b->face_vectors_valid = false;
    b->geom_is_host = 15;
    b->normals_from_x_ok = normals_match_nodes(b);
    return 0;
}

// x only; adflow_gpu_update_geometry derives the rest on the device
int adflow_gpu_upload_coordinates(int nn, int level, int sps)
{
    Block* b = find_block(nn, level, sps);
    if (!b) return fail("block (%d,%d,%d) not registered", nn, level, sps);
    if (!b->d.x) return fail("block (%d,%d,%d): no coordinate array", nn, level, sps);
    if (copy_box(b, b->v.x, b->d.x, 3, 0, b->v.ie + 1, 0, b->v.je + 1, 0, b->v.ke + 1, true)) return 1;
    b->face_vectors_valid = false;
    b->geom_is_host |= 1;
    // the stored sI / sJ / sK are now those of the OLD nodes: until update_geometry (or an upload of the new normals) re-forms
    // them no kernel may take its normals from x while another reads the arrays (two geometries in one residual)
    b->normals_from_x_ok = false;
    return sync_and_check();
}

// volume_block + metric_block + boundaryNormals (adjointExtra.F90:5-364) for every block of the level:
// the `useSpatial` branch of blocketteRes (blockette.F90:203-211) after the mesh moved
int adflow_gpu_update_geometry(int level)
{
    if (need_ready()) return 1;
    int rc = for_level(level, [&](Block* b) {
        if (!b->geom_uploaded) return fail("geometry of a level-%d block has not been uploaded", level);
        launch_volume_metric(b->v, b->d.rightHanded, g_stream);
        if (!b->bc.empty()) launch_boundary_normals(b->v, b->bc.data(), (int)b->bc.size(), g_stream);
        b->face_vectors_valid = false;
        b->geom_is_host &= 1;               // sI / sJ / sK were re-formed on the device: they are no longer the host arrays ...
        b->normals_from_x_ok = true;        // ... but they ARE metric_block(x) of the device's nodes
        return 0;
    });
    if (rc) return rc;
    return sync_and_check();
}

// flowDoms(nn,level,sps)%surfNodeIndices / %uv of determineWallAssociation (wallDistance.F90:1663-2002), kept on the device
int adflow_gpu_wall_distance_register(int nn, int level, int sps, const int32_t* surfNodeIndices, const double* uv)
{
    Block* b = find_block(nn, level, sps);
    if (!b) return fail("block (%d,%d,%d) not registered", nn, level, sps);
    if (!surfNodeIndices || !uv) return fail("wall_distance_register: surfNodeIndices / uv is NULL");
    const size_t n = (size_t)b->v.nx * b->v.ny * b->v.nz;
    // largest surface node any cell refers to: update_wall_distances checks it against the xSurf it is handed
    int32_t mx = 0;
    for (size_t q = 0; q < 4 * n; ++q) {
        if (surfNodeIndices[q] < 0) return fail("wall_distance_register: negative surface node index");
        mx = std::max(mx, surfNodeIndices[q]);
    }
    b->wd_max_node = mx;
    if (!b->wd_ind) {
        void *pi = nullptr, *pu = nullptr;
        HIPCHK(hipMalloc(&pi, n * 4 * sizeof(int32_t)));
        b->allocs.push_back(pi);
        HIPCHK(hipMalloc(&pu, n * 2 * sizeof(double)));
        b->allocs.push_back(pu);
        b->wd_ind = (int*)pi;
        b->wd_uv = (double*)pu;
    }
    HIPCHK(hipMemcpyAsync(b->wd_ind, surfNodeIndices, n * 4 * sizeof(int32_t), hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemcpyAsync(b->wd_uv, uv, n * 2 * sizeof(double), hipMemcpyHostToDevice, g_stream));
    return sync_and_check();
}

static double* g_xsurf = nullptr;
static size_t g_xsurf_n = 0;

static void free_xsurf(void)
{
    if (g_xsurf) (void)hipFree(g_xsurf);
    g_xsurf = nullptr;
    g_xsurf_n = 0;
}

// updateWallDistancesQuickly (wallDistance.F90:36-120) for every block of the level with a registered association
int adflow_gpu_update_wall_distances(int level, const double* xSurf, int64_t n)
{
    if (need_ready()) return 1;
    if (n < 0 || (n > 0 && !xSurf) || n % 3) return fail("update_wall_distances: xSurf with %lld entries", (long long)n);
    if ((size_t)n > g_xsurf_n) {
        if (g_xsurf) (void)hipFree(g_xsurf);
        g_xsurf = nullptr; g_xsurf_n = 0;
        HIPCHK(hipMalloc((void**)&g_xsurf, sizeof(double) * (size_t)n));
        g_xsurf_n = (size_t)n;
    }
    if (n > 0) HIPCHK(hipMemcpyAsync(g_xsurf, xSurf, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, g_stream));
    bool any = false;
    int rc = for_level(level, [&](Block* b) {
        if (!b->wd_ind) return 0;
        if (!b->geom_uploaded) return fail("geometry of a level-%d block has not been uploaded", level);
        if ((int64_t)3 * b->wd_max_node > n)
            return fail("update_wall_distances: the association of a level-%d block refers to surface node %d but xSurf holds %lld nodes",
                        level, b->wd_max_node, (long long)(n / 3));
        any = true;
        launch_wall_distance(b->v, b->wd_ind, b->wd_uv, g_xsurf, g_stream);
        return 0;
    });
    if (rc) return rc;
    (void)any;          // no association on this level (a level without walls): nothing to update, as updateWallDistancesQuickly
    return sync_and_check();
}

int adflow_gpu_upload_state(int nn, int level, int sps)
{
    Block* b = find_block(nn, level, sps);
    if (!b) return fail("block (%d,%d,%d) not registered", nn, level, sps);
    const adflow_block_desc& d = b->d;
    BlkView& v = b->v;
    if (!d.w || !d.p || (!d.gamma && level == 1))
        return fail("block (%d,%d,%d): w, p and (on the finest level) gamma host arrays are required", nn, level, sps);
    int rc = 0;
    if (!d.gamma) {
        // coarse levels of the reference alias the fine level's gamma with fine strides (not handed over): constant gamma
        std::vector<double> g((size_t)(v.ib + 1) * (v.jb + 1) * (v.kb + 1), g_opts.gammaConstant);
        rc |= copy_box(b, v.gamma, g.data(), 1, 0, v.ib + 1, 0, v.jb + 1, 0, v.kb + 1, true);
    }
    rc |= copy_box(b, v.w, d.w, v.nw, 0, v.ib + 1, 0, v.jb + 1, 0, v.kb + 1, true);
    rc |= copy_box(b, v.p, d.p, 1, 0, v.ib + 1, 0, v.jb + 1, 0, v.kb + 1, true);
    rc |= copy_box(b, v.gamma, d.gamma, 1, 0, v.ib + 1, 0, v.jb + 1, 0, v.kb + 1, true);
    rc |= copy_box(b, v.rlv, d.rlv, 1, 0, v.ib + 1, 0, v.jb + 1, 0, v.kb + 1, true);
    rc |= copy_box(b, v.rev, d.rev, 1, 0, v.ib + 1, 0, v.jb + 1, 0, v.kb + 1, true);
    if (rc) return rc;
    b->ss_valid = false;
    b->etot_consistent = false;
    return sync_and_check();
}

int adflow_gpu_download_state(int nn, int level, int sps)
{
    Block* b = find_block(nn, level, sps);
    if (!b) return fail("block (%d,%d,%d) not registered", nn, level, sps);
    const adflow_block_desc& d = b->d;
    BlkView& v = b->v;
    int rc = 0;
    rc |= copy_box(b, v.w, d.w, v.nw, 0, v.ib + 1, 0, v.jb + 1, 0, v.kb + 1, false);
    rc |= copy_box(b, v.p, d.p, 1, 0, v.ib + 1, 0, v.jb + 1, 0, v.kb + 1, false);
    rc |= copy_box(b, v.rlv, d.rlv, 1, 0, v.ib + 1, 0, v.jb + 1, 0, v.kb + 1, false);
    rc |= copy_box(b, v.rev, d.rev, 1, 0, v.ib + 1, 0, v.jb + 1, 0, v.kb + 1, false);
    if (rc) return rc;
    return sync_and_check();
}

int adflow_gpu_download_residual(int nn, int level, int sps)
{
    Block* b = find_block(nn, level, sps);
    if (!b) return fail("block (%d,%d,%d) not registered", nn, level, sps);
    if (!b->d.dw) return fail("block (%d,%d,%d): no host dw array registered", nn, level, sps);
    BlkView& v = b->v;
    int rc = copy_box(b, v.dw, b->d.dw, v.nw, 0, v.ib + 1, 0, v.jb + 1, 0, v.kb + 1, false);
    if (rc) return rc;
    return sync_and_check();
}

static int array_spec(Block* b, int which, double** dev, int* nc, int lo[3], int n[3])
{
    BlkView& v = b->v;
    auto cell = [&]() { lo[0] = lo[1] = lo[2] = 0; n[0] = v.ib + 1; n[1] = v.jb + 1; n[2] = v.kb + 1; };
    auto halo1 = [&]() { lo[0] = lo[1] = lo[2] = 1; n[0] = v.ie; n[1] = v.je; n[2] = v.ke; };
    auto owned = [&]() { lo[0] = lo[1] = lo[2] = 2; n[0] = v.nx; n[1] = v.ny; n[2] = v.nz; };
    *nc = 1;
    switch (which) {
    case ADFLOW_ARR_W: *dev = v.w; *nc = v.nw; cell(); break;
    case ADFLOW_ARR_P: *dev = v.p; cell(); break;
    case ADFLOW_ARR_GAMMA: *dev = v.gamma; cell(); break;
    case ADFLOW_ARR_RLV: *dev = v.rlv; cell(); break;
    case ADFLOW_ARR_REV: *dev = v.rev; cell(); break;
    case ADFLOW_ARR_DW: *dev = v.dw; *nc = v.nw; cell(); break;
    case ADFLOW_ARR_FW: *dev = v.fw; *nc = 5; cell(); break;
    case ADFLOW_ARR_AA: *dev = v.aa; cell(); break;
    case ADFLOW_ARR_VOL: *dev = v.vol; cell(); break;
    case ADFLOW_ARR_DTL: *dev = v.dtl; halo1(); break;
    case ADFLOW_ARR_RADI: *dev = v.radI; halo1(); break;
    case ADFLOW_ARR_RADJ: *dev = v.radJ; halo1(); break;
    case ADFLOW_ARR_RADK: *dev = v.radK; halo1(); break;
    case ADFLOW_ARR_P1: *dev = v.p1; halo1(); break;
    case ADFLOW_ARR_W1: *dev = v.w1; *nc = 5; halo1(); break;
    case ADFLOW_ARR_WN: *dev = v.wn; *nc = 5; owned(); break;
    case ADFLOW_ARR_PN: *dev = v.pn; owned(); break;
    case ADFLOW_ARR_WR: *dev = v.wr; *nc = 5; owned(); break;
    case ADFLOW_ARR_NODAL_GRADS:
        *dev = v.grad; *nc = 12; lo[0] = lo[1] = lo[2] = 1; n[0] = v.il; n[1] = v.jl; n[2] = v.kl; break;
    case ADFLOW_ARR_D2WALL: *dev = v.d2wall; owned(); break;
    case ADFLOW_ARR_X: *dev = v.x; *nc = 3; lo[0] = lo[1] = lo[2] = 0; n[0] = v.ie + 1; n[1] = v.je + 1; n[2] = v.ke + 1; break;
    case ADFLOW_ARR_SI: *dev = v.sI; *nc = 3; lo[0] = 0; lo[1] = lo[2] = 1; n[0] = v.ie + 1; n[1] = v.je; n[2] = v.ke; break;
    case ADFLOW_ARR_SJ: *dev = v.sJ; *nc = 3; lo[1] = 0; lo[0] = lo[2] = 1; n[0] = v.ie; n[1] = v.je + 1; n[2] = v.ke; break;
    case ADFLOW_ARR_SK: *dev = v.sK; *nc = 3; lo[2] = 0; lo[0] = lo[1] = 1; n[0] = v.ie; n[1] = v.je; n[2] = v.ke + 1; break;
    default: return fail("unknown array id %d", which);
    }
    return 0;
}

int adflow_gpu_download_array(int nn, int level, int sps, int which, double* host)
{
    Block* b = find_block(nn, level, sps);
    if (!b) return fail("block (%d,%d,%d) not registered", nn, level, sps);
    if (!host) return fail("null host pointer");
    double* dev; int nc, lo[3], n[3];
    if (array_spec(b, which, &dev, &nc, lo, n)) return 1;
    if (copy_box(b, dev, host, nc, lo[0], n[0], lo[1], n[1], lo[2], n[2], false)) return 1;
    return sync_and_check();
}

int adflow_gpu_upload_array(int nn, int level, int sps, int which, const double* host)
{
    Block* b = find_block(nn, level, sps);
    if (!b) return fail("block (%d,%d,%d) not registered", nn, level, sps);
    if (!host) return fail("null host pointer");
    double* dev; int nc, lo[3], n[3];
    if (array_spec(b, which, &dev, &nc, lo, n)) return 1;
    if (copy_box(b, dev, host, nc, lo[0], n[0], lo[1], n[1], lo[2], n[2], true)) return 1;
    if (which == ADFLOW_ARR_W || which == ADFLOW_ARR_P || which == ADFLOW_ARR_GAMMA) {
        b->ss_valid = false;
        b->etot_consistent = false;
    }
    if (which == ADFLOW_ARR_X || which == ADFLOW_ARR_SI || which == ADFLOW_ARR_SJ || which == ADFLOW_ARR_SK) {
        // nodes or normals replaced one array at a time: the derived face vectors are stale.  The normals count as metric_block(x)
        // again only when the array came from the registered descriptor pointer (then the host arrays the check reads ARE what the
        // device holds); after an upload from any other buffer the stored normals are used until update_geometry / upload_geometry
        // (round-3 advisor finding)
        if (sync_and_check()) return 1;
        b->face_vectors_valid = false;
        const double* own = which == ADFLOW_ARR_X ? b->d.x : which == ADFLOW_ARR_SI ? b->d.sI : which == ADFLOW_ARR_SJ ? b->d.sJ : b->d.sK;
        const unsigned bit = which == ADFLOW_ARR_X ? 1u : which == ADFLOW_ARR_SI ? 2u : which == ADFLOW_ARR_SJ ? 4u : 8u;
        if (host == own) b->geom_is_host |= bit; else b->geom_is_host &= ~bit;
        // the host-side comparison speaks for the device only when ALL FOUR device arrays are the descriptor's: a foreign sI followed by
        // an upload of x from the registered pointer must not switch the re-formed normals back on
        b->normals_from_x_ok = b->geom_is_host == 15 && normals_match_nodes(b);
    }
    return sync_and_check();
}

// ---------------------------------------------------------------- hot path
int adflow_gpu_time_step(int level, int onlyRadii)
{
    if (need_ready()) return 1;
    KParams kp = make_kparams(level, 1.0, 0);
    kp.onlyRadii = onlyRadii;
    if (time_step_level(level, kp)) return 1;
    return sync_and_check();
}

int adflow_gpu_initres(int level, int varStart, int varEnd)
{
    if (need_ready()) return 1;
    if (varEnd < varStart) return 0;
    KParams kp = make_kparams(level, 1.0, 0);
    int rc = for_level(level, [&](Block* b) {
        if (varStart < 1 || varEnd > b->v.nw) return fail("initres: variable range %d..%d outside 1..%d", varStart, varEnd, b->v.nw);
        return 0;
    });
    if (rc) return rc;
    LevelTab t;
    if (level_tab(level, &t)) return 1;
    launch_initres_level(t.tab, t.n, t.nx, t.ny, t.nz, kp, varStart - 1, varEnd - 1, g_stream);
    return sync_and_check();
}

static int enqueue_flow_fluxes(int level, const KParams& kp, bool viscApprox, bool needGrad);
static int source_terms_enqueue(int withBlank);

// residual (residuals.F90:1028) = residual_block of every block; blockResCore (blockette.F90:755) is the same sum of
// fluxes WITHOUT the low-speed preconditioner of residual_block (residuals.F90:172-331) -> lowSpeed = false there.
static int wall_stress_enqueue(int level, const KParams& kp, bool formGrad);

// stage0: the reference's rkStage is 0 at this call -> on the ground level viscousFlux also stores the wall stress tensor
// and heat flux of the viscous subfaces (storeWallTensor, fluxes.F90:2586-2592)
static bool has_wall_subfaces(int level);

// needGradHbm: the caller wants the nodal gradients in the block arrays (updateIntermed copy-out, blockette.F90:706-750)
// the scheme of the preconditioner matrix that k_pc_march serves: first-order upwind (lumpedDiss, or the user's first-order limiter) on
// the fine level, no matrix-free vector, no multigrid forcing
static bool pc_march_scheme(const KParams& kp)
{
    const int lim = kp.lumpedDiss ? ADFLOW_LIM_FIRST_ORDER : kp.limiter;
    return g_pc_fused && kp.spaceDiscr == ADFLOW_UPWIND && kp.fineGrid && lim == ADFLOW_LIM_FIRST_ORDER && !kp.rvec && !kp.coarseInit &&
           roe_march_takes(kp);
}
// ... and the conditions under which enqueue_flow_residual reaches it (thin-layer viscous flux, blocks at rest, the marching kernels on)
static bool pc_march_applies(int level, const KParams& kp, bool viscApprox)
{
    bool anyMoving = false;
    for_level(level, [&](Block* b) { anyMoving = anyMoving || b->v.sFace || b->v.moving; return 0; });
    return viscApprox && viscous_is_tiled() >= 2 && kp.viscous && fabs(kp.rFil) >= 1.e-10 && !kp.fwMode && inviscid_march_enabled() &&
           !anyMoving && pc_march_scheme(kp);
}
// the exact viscous residual on the upwind scheme ends in k_roe_march<.., ADDV, RV> when a matrix-free vector is the target
// (enqueue_flow_residual: viscFirst): that kernel can write all six entries of a cell (KParams::rvecTurbFromDw)
static bool roe_rv_completes(int level, const KParams& kp, bool viscApprox)
{
    bool anyMoving = false;
    for_level(level, [&](Block* b) { anyMoving = anyMoving || b->v.sFace || b->v.moving; return 0; });
    return kp.rvec && kp.viscous && fabs(kp.rFil) >= 1.e-10 && !viscApprox && viscous_is_tiled() >= 2 && !kp.fwMode && inviscid_march_enabled() &&
           !kp.dissApprox && !kp.lumpedDiss && !anyMoving && roe_march_takes(kp);
}
static bool ad_pc_march_applies(const KParams& kp, bool viscApprox)
{
    return viscApprox && viscous_is_tiled() >= 2 && kp.viscous && fabs(kp.rFil) >= 1.e-10 && pc_march_scheme(kp);
}

static int enqueue_flow_residual(int level, const KParams& kp, bool viscApprox = false, bool lowSpeed = true, bool stage0 = true,
                                 bool needGradHbm = false)
{
    const bool wallStress = stage0 && !viscApprox && kp.viscous && level == g_opts.groundLevel && fabs(kp.rFil) >= 1.e-10 &&
                            has_wall_subfaces(level);
    // the wall stress reads the gradients of the node planes ON the wall faces only: when the flux kernel keeps its gradients in
    // LDS (k_visc_gf) those few nodes are formed again by the wall-stress launch instead of every node of the level being stored
    const bool gradOnChip = viscous_is_tiled() >= 2 && !needGradHbm;
    if (enqueue_flow_fluxes(level, kp, viscApprox, needGradHbm)) return 1;
    if (wallStress)
        if (wall_stress_enqueue(level, kp, gradOnChip)) return 1;
    // sourceTerms() of the call sites of `residual` (smoothers.F90:74,409, multiGrid.F90:52,887,949): fine level only
    if (lowSpeed && level == 1 && source_terms_enqueue(1)) return 1;
    if (lowSpeed && g_opts.lowSpeedPreconditioner) {
        LevelTab t;
        if (level_tab(level, &t)) return 1;
        launch_low_speed_precond_level(t.tab, t.n, t.nx, t.ny, t.nz, kp, g_stream);
    }
    return 0;
}

static int enqueue_flow_fluxes(int level, const KParams& kp, bool viscApprox, bool needGrad)
{
    bool anyMoving = false;     // grid velocities / rotational source: the generic kernels carry them
    for_level(level, [&](Block* b) { anyMoving = anyMoving || b->v.sFace || b->v.moving; return 0; });
    if (g_use_march && !kp.viscous && kp.spaceDiscr == ADFLOW_DISS_SCALAR && kp.fineGrid && !kp.dissApprox && !anyMoving) {
        // Euler + scalar JST: one k-marching launch over every block of the level
        int rc = for_level(level, [&](Block* b) {
            if (!b->geom_uploaded) return fail("geometry of a level-%d block has not been uploaded", level);
            return 0;
        });
        if (rc) return rc;
        if (ensure_tiles(level)) return 1;
        launch_euler_march(g_tab[level], g_tiles[level].first, g_tiles[level].second, kp, g_stream);
        return 0;
    }
    if (kp.spaceDiscr != ADFLOW_DISS_SCALAR && kp.spaceDiscr != ADFLOW_DISS_MATRIX && kp.spaceDiscr != ADFLOW_UPWIND)
        return fail("spaceDiscr=%d not supported (1 scalar, 2 matrix, 9 upwind)", kp.spaceDiscr);
    const bool wantSensor = kp.viscous && kp.spaceDiscr == ADFLOW_DISS_SCALAR && fabs(kp.rFil) >= 1.e-10 && !kp.dissApprox;
    int nStale = 0, nBlk = 0;
    int rc = for_level(level, [&](Block* b) {
        if (!b->geom_uploaded) return fail("geometry of a level-%d block has not been uploaded", level);
        ++nBlk;
        if (wantSensor && !b->ss_valid) ++nStale;
        return 0;
    });
    if (rc) return rc;
    if (nStale > 0) {
        // entropy sensor of the blocks whose state changed: one launch when that is every block of the level (the usual case)
        if (nStale == nBlk) {
            LevelTab ts;
            if (level_tab(level, &ts)) return 1;
            launch_entropy_level(ts.tab, ts.n, ts.nx, ts.ny, ts.nz, g_stream);
        }
        for_level(level, [&](Block* b) {
            if (nStale != nBlk && !b->ss_valid) launch_entropy(b->v, g_stream);
            b->ss_valid = true;
            return 0;
        });
    }
    // inviscid part: one launch for every block of the level (blocks are independent given their halos)
    LevelTab t;
    if (level_tab(level, &t)) return 1;
    // exact viscous fluxes of blocks at rest: nodal gradients + face fluxes as ONE marching kernel (k_visc_gf)
    const bool viscMarch = kp.viscous && fabs(kp.rFil) >= 1.e-10 && !viscApprox && viscous_is_tiled() >= 2;
    // viscous march first, inviscid march last: the inviscid kernel (Roe: bound by FP64 issue) adds the viscous sums it finds in
    // dw(2:5) instead of the viscous kernel reading dw back.  Any inviscid kernel over the tile table can take that role (Roe,
    // matrix dissipation, scalar JST of NS / RANS); not with the persistent fw of the Runge-Kutta stages.
    // (scalar JST: the marching form reads its sensor from b.ss -- the entropy sensor of NS / RANS, or the frozen sensor of the
    //  approximate residual, which Euler has too)
    const bool scalarViscM = (inviscid_march_enabled() >= 2 && kp.spaceDiscr == ADFLOW_DISS_SCALAR && (kp.viscous || kp.dissApprox) &&
                              kp.fineGrid);
    // the approximate residual of the preconditioner matrix: the lumped scalar / matrix dissipation has a marching form on the fine
    // level (k_inviscid_march<.., APX>, round 6); the upwind scheme changes through its limiter only
    const bool approxOk = !kp.dissApprox || kp.spaceDiscr == ADFLOW_UPWIND || (kp.fineGrid && g_pc_fused && !kp.fwMode);
    const bool tileInviscid = inviscid_march_enabled() && (kp.spaceDiscr != ADFLOW_DISS_SCALAR || scalarViscM) && approxOk && !anyMoving;
    const bool viscFirst = viscMarch && !kp.fwMode && tileInviscid && !kp.dissApprox && !kp.lumpedDiss;
    // the same order for the thin-layer viscous march of the preconditioner assembly (no gradients)
    const bool approxFirst = viscApprox && viscous_is_tiled() >= 2 && kp.viscous && fabs(kp.rFil) >= 1.e-10 && !kp.fwMode &&
                             tileInviscid;
    // scalar JST with the entropy sensor (NS / RANS, fine level) also has a marching form, but it is bound by memory like
    // the gather form (1.06 vs 1.10 ms on 8 x 128x128x96): only with tuning inviscid_march = 2
    if (viscFirst || approxFirst) {
        // enqueued behind the viscous march below
    } else if (tileInviscid) {
        // (the approximate residual changes the Roe scheme only through the limiter: inviscidUpwindFlux is called either way)
        // matrix dissipation / Roe upwind: k-marching kernel over the level's tile table (every face once in k and i)
        if (ensure_tiles(level)) return 1;
        // second-order Roe upwind on the fine level: the per-cell reconstruction kernel; everything else (matrix, first order)
        // stays with the per-face kernel
        if (!launch_roe_march(g_tab[level], g_tiles[level].first, g_tiles[level].second, kp, g_stream))
            launch_inviscid_march(g_tab[level], g_tiles[level].first, g_tiles[level].second, kp, g_stream);
    } else {
        launch_inviscid_level(t.tab, t.n, t.nx, t.ny, t.nz, kp, g_stream);
    }
    if (!viscFirst && !approxFirst) phase_mark(4);
    if (!(kp.viscous && fabs(kp.rFil) >= 1.e-10)) return 0;
    const bool batched = !viscApprox && viscMarch;
    // thin-layer viscous flux of the preconditioner assembly: marching form over the tile table (blocks at rest, 4-row tiles)
    const bool approxMarch = viscApprox && viscous_is_tiled() >= 2 && !anyMoving;
    rc = for_level(level, [&](Block* b) {
        if (!b->face_vectors_valid) {
            launch_face_vectors(b->v, g_stream);
            b->face_vectors_valid = true;
        }
        if (viscApprox && !approxMarch) launch_viscous_approx(b->v, kp, g_stream);   // viscousFluxApprox instead of gradients + viscousFlux
        else if (!viscApprox && !batched) launch_viscous(b->v, kp, g_stream);
        return 0;
    });
    if (rc) return rc;
    if (approxMarch) {
        if (ensure_tiles(level)) return 1;
        if (approxFirst && pc_march_scheme(kp)) {
            // first-order upwind + thin-layer viscous flux: both are functions of the two cells of a face -- one march, dw written once
            phase_mark(4);
            launch_pc_march(g_tab[level], g_tiles[level].first, g_tiles[level].second, kp, g_march_kch, g_stream);
            phase_mark(5);
        } else if (approxFirst) {
            KParams kv = kp;
            kv.viscFirst = 1;
            phase_mark(4);
            launch_visc_march_approx(g_tab[level], g_tiles[level].first, g_tiles[level].second, kv, g_stream);
            phase_mark(5);
            if (!launch_roe_march(g_tab[level], g_tiles[level].first, g_tiles[level].second, kv, g_stream))
                launch_inviscid_march(g_tab[level], g_tiles[level].first, g_tiles[level].second, kv, g_stream);
        } else
            launch_visc_march_approx(g_tab[level], g_tiles[level].first, g_tiles[level].second, kp, g_stream);
    }
    if (batched) {
        // gradients and face fluxes in one kernel: the gradients stay in LDS (and go to HBM only when a caller reads them)
        // (phase marks: 4 .. 5 = nodal gradients + viscous fluxes, 5 .. 6 = inviscid fluxes when they follow; bench.py labels them so)
        if (ensure_gf_tiles(level)) return 1;
        if (viscFirst) {
            KParams kv = kp;
            kv.viscFirst = 1;
            if (ensure_tiles(level)) return 1;
            phase_mark(4);
            launch_visc_gf(g_tab[level], g_gf_tiles[level].first, g_gf_tiles[level].second, kv, needGrad, g_stream);
            phase_mark(5);
            if (!launch_roe_march(g_tab[level], g_tiles[level].first, g_tiles[level].second, kv, g_stream))
                launch_inviscid_march(g_tab[level], g_tiles[level].first, g_tiles[level].second, kv, g_stream);
            return 0;
        }
        phase_mark(5);
        launch_visc_gf(g_tab[level], g_gf_tiles[level].first, g_gf_tiles[level].second, kp, needGrad, g_stream);
    }
    return 0;
}

int adflow_gpu_residual(int level, int rkStage)
{
    if (need_ready()) return 1;
    double rFil = 1.0;
    int fwMode = 0;
    if (g_opts.smoother == ADFLOW_RUNGE_KUTTA) {
        if (rkStage < 0 || rkStage >= g_opts.nRKStages) return fail("rkStage=%d outside 0..%d", rkStage, g_opts.nRKStages - 1);
        rFil = g_opts.cdisRK[rkStage];   // cdisRK(rkStage+1), residuals.F90:61-65
        fwMode = 1;
    }
    KParams kp = make_kparams(level, rFil, fwMode);
    int rc = enqueue_flow_residual(level, kp, false, true, rkStage == 0);
    if (rc) return rc;
    return sync_and_check();
}

static int block_res_enqueue(int level, unsigned flags);
static int apply_bc_enqueue(int level, int secondHalo);
static int apply_turb_bc_enqueue(int level, int secondHalo);
static int apply_turb_and_flow_bc_enqueue(int level, int secondHalo, bool turbBC);
static int turb_bc_treatment_enqueue(int level, const KParams& kp);
static int turb_bc_apply_enqueue(int level, const KParams& kp, int secondHalo);
static int bc_coarse_corrections_enqueue(int coarseLevel, double fact);
static void bc_plan_drop(int level);
static int halo_exchange_enqueue(int level, int varStart, int varEnd, int commPressure, int commVisc, int nLayers);
static int early_pressure_exchange_enqueue(int level);
static int halo_exchange_close(int level, int varStart, int varEnd, int commPressure, int nLayers);
static int comm_exchange_begin(CommPattern* cp, BlkView* tab, unsigned mask, int nvar, bool* remoteOut);
static int comm_exchange_end(CommPattern* cp, BlkView* tab, unsigned mask, bool remote);
static int block_res_split_enqueue(int level, unsigned flags, const KParams& kp0, int lStart, int lEnd, int* taken,
                                   const std::function<int()>& frontBCs);

// whalo2 + blocketteRes core with the exchange -- and, round 4, the boundary conditions in front of it -- HIDDEN behind the tiles that
// read no halo cell (round-2 verdict, next 3 iii).  Two queues from the fork behind the derived values:
//   main queue:  boundary conditions (frontBCs), packs, RCCL group on the communication queue, same-GPU copies, wait for the group,
//                unpacks, periodic transforms, whalo2's closing energy; then the fused viscous march over the BOUNDARY tiles, the
//                inviscid march over every tile (it adds the viscous sums), the wall stress
//   side queue:  SA march and fused viscous march over their INTERIOR tiles (they read owned cells only: neither a boundary halo
//                nor an exchanged one) -- while the boundary kernels (a few hundred workgroups each) and the copies leave the chip
//                idle; then, behind the exchange, the SA march over the boundary tiles
// Taken for the default flags of blocketteRes on NS / RANS with the marching kernels, blocks at rest, when the pattern has messages
// (tuning "split_eval" = 1, the default), always (2: tests), never (0).  On the north-star mesh
// 16 % of the tiles are interior.
static int block_res_split_enqueue(int level, unsigned flags, const KParams& kp0, int lStart, int lEnd, int* taken,
                                   const std::function<int()>& frontBCs)
{
    *taken = 0;
    if (!g_split_eval || !g_overlap || g_phase_base > 0 || kp0.rvec) return 0;
    const unsigned need = ADFLOW_RES_FLOW | ADFLOW_RES_HALO;
    if ((flags & need) != need || (flags & (ADFLOW_RES_DISS_APPROX | ADFLOW_RES_VISC_APPROX | ADFLOW_RES_UPDATE_INTERMED))) return 0;
    KParams kp = kp0;
    if (!kp.viscous || fabs(kp.rFil) < 1.e-10 || viscous_is_tiled() < 2 || !inviscid_march_enabled() || !kp.fineGrid) return 0;
    if (kp.spaceDiscr == ADFLOW_DISS_SCALAR) return 0;         // (needs the time-step pass in front: not split)
    const bool rans = (flags & ADFLOW_RES_TURB) && g_opts.equations == ADFLOW_RANS;
    if (rans && !g_sa_march) return 0;
    bool moving = false, wall = has_wall_subfaces(level) && level == g_opts.groundLevel;
    for_level(level, [&](Block* b) { moving = moving || b->v.sFace || b->v.moving; return 0; });
    if (moving || !g_act.empty()) return 0;
    CommPattern* cp;
    if (build_comm(level, 2, &cp)) return 1;
    const bool messages = !cp->sends.empty() || !cp->recvs.empty();
    // (boundary subfaces alone do not make it pay: measured on the wall-bounded bench mesh at N = 1 the split costs 0.03 ms --
    //  2.45 against 2.42 ms -- the two extra launches per kernel outweigh what the interior tiles hide of the boundary kernels)
    if (g_split_eval < 2 && !messages) return 0;
    if (g_bc_callback) return 0;               // (a host hook between the device passes synchronises the queue)
    unsigned mask; int nvar;
    if (halo_mask(lStart, lEnd, 1, 1, &mask, &nvar)) return 1;
    if (nvar == 0) return 0;
    LevelTab t;
    if (level_tab(level, &t)) return 1;
    int rc = for_level(level, [&](Block* b) {
        if (!b->geom_uploaded) return fail("geometry of a level-%d block has not been uploaded", level);
        if (rans && b->v.nw < 6) return fail("RANS/SA needs nw = 6 (block has %d)", b->v.nw);
        if (!b->face_vectors_valid) {
            launch_face_vectors(b->v, g_stream);
            b->face_vectors_valid = true;
        }
        return 0;
    });
    if (rc) return rc;
    if (ensure_tiles(level) || ensure_gf_tiles(level) || (rans && ensure_sa_tiles(level))) return 1;
    *taken = 1;
    KParams kv = kp;
    kv.viscFirst = 1;
    // (everything behind the fork runs inside `run`: an error exit must not leave the side queue unjoined -- round-3 advisor finding)
    bool forked = false;
    auto run = [&]() -> int {
        // ---- fork behind the derived values: the halo-free tiles on the side queue
        HIPCHK(hipEventRecord(g_evFork, g_stream));
        HIPCHK(hipStreamWaitEvent(g_streamB, g_evFork, 0));
        forked = true;
        if (rans) launch_sa_march(t.tab, g_sa_tiles_int[level].first, g_sa_tiles_int[level].second, kp, g_streamB, false);
        launch_visc_gf(g_tab[level], g_gf_tiles_int[level].first, g_gf_tiles_int[level].second, kv, false, g_streamB);
        HIPCHK(hipEventRecord(g_evB1, g_streamB));
        // ---- main queue: boundary conditions, messages out, same-GPU copies, messages in
        if (frontBCs()) return 1;
        if (g_test_fault & 2) return fail("test_fault: the split evaluation fails behind its fork");
        bool remote = false;
        if (comm_exchange_begin(cp, g_tab[level], mask, nvar, &remote)) return 1;
        if (comm_exchange_end(cp, g_tab[level], mask, remote)) return 1;
        if (halo_exchange_close(level, lStart, lEnd, 1, 2)) return 1;
        // ---- the tiles next to the block faces, then the inviscid march over all of them
        HIPCHK(hipEventRecord(g_evC, g_stream));
        HIPCHK(hipStreamWaitEvent(g_streamB, g_evC, 0));
        if (rans) launch_sa_march(t.tab, g_sa_tiles_bnd[level].first, g_sa_tiles_bnd[level].second, kp, g_streamB, false);
        HIPCHK(hipEventRecord(g_evB, g_streamB));
        launch_visc_gf(g_tab[level], g_gf_tiles_bnd[level].first, g_gf_tiles_bnd[level].second, kv, false, g_stream);
        HIPCHK(hipStreamWaitEvent(g_stream, g_evB1, 0));               // the interior viscous sums are in dw(2:5)
        if (!launch_roe_march(g_tab[level], g_tiles[level].first, g_tiles[level].second, kv, g_stream))
            launch_inviscid_march(g_tab[level], g_tiles[level].first, g_tiles[level].second, kv, g_stream);
        HIPCHK(hipStreamWaitEvent(g_stream, g_evB, 0));                // join
        forked = false;
        return 0;
    };
    if (run()) {
        if (forked && hipEventRecord(g_evB, g_streamB) == hipSuccess) (void)hipStreamWaitEvent(g_stream, g_evB, 0);
        return 1;
    }
    // viscSubface%tau / %q (storeWall, blockette.F90:135): from the node planes on the wall faces, formed behind the exchange
    if (wall && wall_stress_enqueue(level, kp, true)) return 1;
    return 0;
}

static int block_res_enqueue(int level, unsigned flags)
{
    if (need_ready()) return 1;
    phase_mark(0);
    KParams kp = make_kparams(level, 1.0, 0);
    kp.onlyRadii = !(flags & ADFLOW_RES_UPDATE_INTERMED);
    kp.coarseInit = 0;
    if (level == 1 && g_rvec_target) { kp.rvec = g_rvec_target; kp.rvecTurbScale = g_opts.turbResScale; }
    kp.dissApprox = (flags & ADFLOW_RES_DISS_APPROX) ? 1 : 0;
    if (kp.dissApprox && (flags & ADFLOW_RES_UPWIND_FIRST_ORDER)) kp.lumpedDiss = 1;   // blockette.F90:643
    const bool viscApprox = (flags & ADFLOW_RES_VISC_APPROX) != 0;
    int rc = 0;
    bool etotInClosures = false;
    if (flags & ADFLOW_RES_CLOSURES) {
        // computePressureSimple / computeLamViscosity / computeEddyViscosity (blockette.F90:199-203)
        LevelTab tc;
        if (level_tab(level, &tc)) return 1;
        // when the whalo2 of this call follows (mean-flow variables incl. the energy), the energy it would recompute on the owned
        // cells is written by the same pass wherever the pressure kept its value (as FormFunction_mf does, kernels_nk.hip
        // closures_body): a floored cell raises the device flag and whalo2's closing pass runs only then.  (The reference sends the old
        // energy and recomputes afterwards, haloExchange.F90:154-196: a neighbour's halo energy differs by the rounding of p -> E -> p,
        // DESIGN 4 round 4 (c))
        etotInClosures = (flags & ADFLOW_RES_HALO) && (flags & ADFLOW_RES_FLOW) && g_comm.count(std::make_pair(level, 2)) > 0;
        if (etotInClosures) {
            if (!g_floor_flag_dev) HIPCHK(hipMalloc((void**)&g_floor_flag_dev, sizeof(int)));
            HIPCHK(hipMemsetAsync(g_floor_flag_dev, 0, sizeof(int), g_stream));
        }
        launch_closures_level(tc.tab, tc.n, tc.nx, tc.ny, tc.nz, kp, g_stream, etotInClosures ? g_floor_flag_dev : nullptr);
        rc = for_level(level, [&](Block* b) {
            b->ss_valid = false;
            b->etot_consistent = false;
            return 0;
        });
        if (rc) return rc;
    }
    if (flags & ADFLOW_RES_HALO) {
        // BCTurbTreatment + applyAllTurbBCThisBlock(.true.) before applyAllBC_block(.true.) (blockette.F90:220-226)
        auto frontBCs = [&]() -> int {
            if (apply_turb_and_flow_bc_enqueue(level, 1, (flags & ADFLOW_RES_TURB) != 0)) return 1;
            if (g_bc_callback) {
                HIPCHK(hipStreamSynchronize(g_stream));
                g_bc_callback(level, 1);
            }
            return 0;
        };
        int lStart = 1, lEnd = (g_opts.equations == ADFLOW_RANS) ? 6 : 5;
        if ((flags & ADFLOW_RES_FLOW) && !(flags & ADFLOW_RES_TURB)) lEnd = 5;
        if (!(flags & ADFLOW_RES_FLOW) && (flags & ADFLOW_RES_TURB)) lStart = 6;
        if (g_comm.count(std::make_pair(level, 2))) {
            // the boundary conditions and the exchange with the halo-free tiles of the evaluation beside them, where the evaluation
            // is the marching RANS / NS one
            int taken = 0;
            const int flagLevelWas = g_etot_flag_level;
            if (etotInClosures) g_etot_flag_level = level;
            rc = block_res_split_enqueue(level, flags, kp, lStart, lEnd, &taken, frontBCs);
            if (!rc && !taken) {
                rc = frontBCs();
                if (!rc) rc = halo_exchange_enqueue(level, lStart, lEnd, 1, 1, 2);
            }
            g_etot_flag_level = flagLevelWas;
            if (rc) return 1;
            if (taken) return 0;
        } else if (frontBCs()) return 1;
    }
    phase_mark(1);
    // timeStep_block(onlyRadii): with matrix dissipation / Roe upwind nothing in the residual reads the spectral radii, and
    // without updateIntermed the reference's default path (blocketteResCore, blockette.F90:299-753) keeps them in the
    // blockette's private arrays: they are not an output of the evaluation.  Only scalar JST needs them (and the entropy
    // sensor the same kernel leaves in ss).
    // Euler + scalar JST: the marching kernel forms the radii itself when nothing else needs them (no updateIntermed: neither the
    // radii nor dtl are outputs, blockette.F90:660-750)
    bool anyMovingR = false;
    for_level(level, [&](Block* b) { anyMovingR = anyMovingR || b->v.sFace || b->v.moving; return 0; });
    const bool radiiInMarch = kp.onlyRadii && (flags & ADFLOW_RES_FLOW) && g_use_march && !kp.viscous &&
                              kp.spaceDiscr == ADFLOW_DISS_SCALAR && kp.fineGrid && !kp.dissApprox && !anyMovingR && fabs(kp.rFil) >= 1.e-10 &&
                              euler_march_radii_capable(kp);
    kp.radiiInMarch = radiiInMarch ? 1 : 0;
    if (!radiiInMarch && !(kp.onlyRadii && kp.spaceDiscr != ADFLOW_DISS_SCALAR)) {
        rc = time_step_level(level, kp);
        if (rc) return rc;
    }
    phase_mark(2);
    // blockResCore order: SA residual first, then the mean-flow fluxes (blockette.F90:806-851)
    if ((flags & ADFLOW_RES_TURB) && g_opts.equations == ADFLOW_RANS) {
        bool moving = false;
        rc = for_level(level, [&](Block* b) {
            if (b->v.nw < 6) return fail("RANS/SA needs nw = 6 (block has %d)", b->v.nw);
            moving = moving || b->v.sFace || b->v.moving;
            return 0;
        });
        if (rc) return rc;
        {
            LevelTab t;
            if (level_tab(level, &t)) return 1;
            // (on a side queue beside the mean-flow kernels -- the default of rounds 2-3 -- the march gains nothing since every march fills
            // the device: 2.26 against 2.21 ms, round 4)
            hipStream_t ss = g_stream;
            if (g_sa_march && !moving) {
                if (ensure_sa_tiles(level)) return 1;
                // the matrix-free vector: the Roe march that follows in the same queue writes the turbulence entry with its own five
                // (tuning "rvec_joint"; every block holds six variables here: checked above)
                if (kp.rvec && g_rvec_joint && (flags & ADFLOW_RES_FLOW) && roe_rv_completes(level, kp, viscApprox)) kp.rvecTurbFromDw = 1;
                launch_sa_march(t.tab, g_sa_tiles[level].first, g_sa_tiles[level].second, kp, ss, false);
            } else launch_sa_residual_level(t.tab, t.n, t.nx, t.ny, t.nz, kp, ss);
        }
    }
    phase_mark(3);
    if (flags & ADFLOW_RES_FLOW) {
        rc = enqueue_flow_residual(level, kp, viscApprox, false, true, (flags & ADFLOW_RES_UPDATE_INTERMED) != 0);
        if (rc) return rc;
    }
    phase_mark(6);
    // actuator-region sources after the core, fine level only and without the iblank factor (blockette.F90:276-281)
    if (level == 1 && source_terms_enqueue(0)) return 1;
    return 0;
}

// ---- actuator regions (actuatorRegionData.F90) -----------------------------------------------------------------------
int adflow_gpu_actuator_register(int nRegions, const adflow_actuator_region* regions)
{
    ++g_state_gen;
    if (g_device < 0) return fail("adflow_gpu_init has not been called");
    if (nRegions < 0 || (nRegions > 0 && !regions)) return fail("actuator_register: nRegions=%d", nRegions);
    if (g_stream) (void)hipStreamSynchronize(g_stream);
    for (auto& r : g_act) free_list(r.list);
    g_act.clear();
    for (int m = 0; m < nRegions; ++m) {
        const adflow_actuator_region& in = regions[m];
        const int n = in.nCellIDs;
        if (n < 0 || (n > 0 && (!in.block || !in.cellIDs))) return fail("actuator_register: region %d: nCellIDs=%d", m + 1, n);
        if (!(in.volume > 0.0)) return fail("actuator_register: region %d: volume must be positive", m + 1);
        ActRegion r;
        r.h_block.assign(in.block, in.block + n);
        r.h_idx.resize((size_t)3 * n);
        for (int t = 0; t < n; ++t)
            for (int q = 0; q < 3; ++q) r.h_idx[(size_t)q * n + t] = in.cellIDs[(size_t)3 * t + q];   // (3,n) -> (n,3)
        for (int q = 0; q < 3; ++q) r.force[q] = in.force[q];
        r.heat = in.heat; r.volume = in.volume; r.relaxStart = in.relaxStart; r.relaxEnd = in.relaxEnd;
        g_act.push_back(r);
    }
    return 0;
}

static int source_terms_enqueue(int withBlank)
{
    if (g_act.empty()) return 0;
    if (ensure_table(1)) return 1;
    for (auto& r : g_act) {
        const int n = (int)r.h_block.size();
        if (!r.built) {
            free_list(r.list);
            r.list.n = n;
            if (n > 0 && make_list(1, r.h_block.data(), r.h_idx.data(), n, 0, n, &r.list.blkA, &r.list.offA)) return 1;
            r.built = true;
        }
        // relaxation factor and the non-dimensional scales (residuals.F90:367-385)
        double factor;
        const double oc = g_opts.ordersConverged;
        if (oc < r.relaxStart) factor = 0.0;
        else if (oc > r.relaxEnd) factor = 1.0;
        else factor = (oc - r.relaxStart) / (r.relaxEnd - r.relaxStart);
        double Ff[3];
        for (int q = 0; q < 3; ++q) Ff[q] = factor * r.force[q] / r.volume / g_opts.pRef;
        const double Qf = factor * r.heat / r.volume / (g_opts.pRef * g_opts.uRef * g_opts.LRef * g_opts.LRef);
        launch_source_terms(g_tab[1], r.list.blkA, r.list.offA, n, Ff, Qf, withBlank, g_stream);
    }
    return 0;
}

// referenceShockSensor (adjointUtils.F90:1909-1969): pressure (Euler or matrix dissipation) or entropy
int adflow_gpu_reference_shock_sensor(int level)
{
    if (need_ready()) return 1;
    const bool pressure = (g_opts.equations == ADFLOW_EULER) || (g_opts.spaceDiscr == ADFLOW_DISS_MATRIX);
    int rc = for_level(level, [&](Block* b) {
        if (pressure) {
            HIPCHK(hipMemcpyAsync(b->v.ss, b->v.p, sizeof(double) * (size_t)b->boxsize, hipMemcpyDeviceToDevice, g_stream));
        } else {
            launch_entropy(b->v, g_stream);
        }
        b->ss_valid = false;    // the exact viscous kernel recomputes its own sensor after an approximate pass
        return 0;
    });
    if (rc) return rc;
    return sync_and_check();
}

int adflow_gpu_block_res(int level, unsigned flags)
{
    int rc = block_res_enqueue(level, flags);
    if (rc) return rc;
    return sync_and_check();
}

// ---- coloured finite-difference Jacobian (adjointUtils::setupStateResidualMatrix, useAD = F; adjointUtils.F90:7-715) ----------
static JacSpec g_jac;                 // stencil / colouring / state range of the last assembly
static bool g_jac_valid = false;

static void jac_spec(unsigned flags, bool viscous, bool rans, JacSpec* J)
{
    memset(J, 0, sizeof *J);
    // state range (adjointUtils.F90:85-106)
    if (flags & ADFLOW_JAC_TURB_ONLY) { J->lStart = 5; J->nState = 1; }
    else { J->lStart = 0; J->nState = (rans && !(flags & ADFLOW_JAC_FROZEN_TURB)) ? 6 : 5; }
    int n = 0;
    auto add = [&](int a, int b, int c) { J->st[n][0] = a; J->st[n][1] = b; J->st[n][2] = c; ++n; };
    auto star = [&](int r) { add(-r, 0, 0); add(r, 0, 0); add(0, -r, 0); add(0, r, 0); add(0, 0, -r); add(0, 0, r); };
    if (flags & ADFLOW_JAC_PC) {
        add(0, 0, 0); star(1);                                          // euler_pc_stencil (stencils.f90:34-40)
        if (viscous && (flags & ADFLOW_JAC_VISC_PC)) {                  // visc_pc_stencil (:60-85), setup_3x3x3_coloring
            for (int c = -1; c <= 1; ++c)
                for (int b = -1; b <= 1; ++b)
                    for (int a = -1; a <= 1; ++a)
                        if ((a != 0) + (b != 0) + (c != 0) >= 2) add(a, b, c);
            J->ca = 1; J->cb = 3; J->cc = 9; J->cn = 27; J->cm = 3;
        } else {
            J->ca = 1; J->cb = 5; J->cc = 4; J->cn = 7; J->cm = 7;      // setup_PC_coloring (adjointUtils.F90:1089-1119)
        }
    } else if (viscous) {
        for (int c = -1; c <= 1; ++c)                                   // visc_drdw_stencil (stencils.f90:87-106)
            for (int b = -1; b <= 1; ++b)
                for (int a = -1; a <= 1; ++a) add(a, b, c);
        star(2);
        J->ca = 1; J->cb = 19; J->cc = 11; J->cn = 35; J->cm = 35;      // setup_dRdw_visc_coloring (:1153-1183)
    } else {
        add(0, 0, 0);                                                   // euler_drdw_stencil (stencils.f90:42-55)
        for (int d = 0; d < 3; ++d)
            for (int r : {-2, -1, 1, 2}) add(d == 0 ? r : 0, d == 1 ? r : 0, d == 2 ? r : 0);
        J->ca = 1; J->cb = 3; J->cc = 4; J->cn = 13; J->cm = 13;        // setup_dRdw_euler_coloring (:1121-1151)
    }
    J->nStencil = n;
    for (int q = 0; q < n; ++q) {
        const int v = (J->ca * J->st[q][0] + J->cb * J->st[q][1] + J->cc * J->st[q][2]) % J->cn;
        J->sc[q] = (v < 0) ? v + J->cn : v;
    }
}

// masterRoutines::block_res_state (masterRoutines.F90:1214-1283) for every block of the level: closures with halos, turbulence
// and mean-flow boundary conditions, the residual core, actuator sources.  resScale is applied by the extraction kernel.
static int block_res_state_enqueue(int level, unsigned resFlags, bool turbBC, bool closuresDone = false)
{
    KParams kp = make_kparams(level, 1.0, 0);
    int rc = for_level(level, [&](Block* b) {
        if (!closuresDone) launch_closures_halo(b->v, kp, g_stream);
        b->ss_valid = false;
        b->etot_consistent = false;
        return 0;
    });
    if (rc) return rc;
    if (apply_turb_and_flow_bc_enqueue(level, 1, turbBC)) return 1;
    if (g_bc_callback) {
        HIPCHK(hipStreamSynchronize(g_stream));
        g_bc_callback(level, 1);
    }
    return block_res_enqueue(level, resFlags);
}

// ---- forward-mode linearisation (kernels_ad.hip): dual copies of the arrays the gather kernels touch ----------------------------
// Round 5 (round-4 advisor): ONE slab per level instead of ~57 hipMalloc / hipFree per block and call, kept between calls (tuning
// "ad_cache", default 1; dropped when a block is released or the level's layout changes) -- on the north-star mesh the 13.5 GB (8.4 GB since the geometry is no longer copied) of
// dual arrays cost 0-400 ms per assembly to map, depending on the box (profiles/r05_f_bench.json against r05_d) -- and the free
// memory is checked before the slab is requested.
struct AdInit { char* dst; const double* src; size_t n, zeroBytes; };   // dual array at dst: (src, 0) for n entries, or zeroBytes of zero
struct AdBlock { BlkView v; std::vector<AdInit> init; };
static std::map<Block*, AdBlock> g_ad;
static BlkView* g_ad_tab = nullptr;        // device table of the level being linearised, slot layout of g_tab[level]
static char* g_ad_slab = nullptr;
static size_t g_ad_slab_bytes = 0, g_ad_bump = 0;
static bool g_ad_measure = false;
static std::vector<long> g_ad_sig;         // what the cached slab was laid out for
int g_ad_cache = 1;

static void ad_drop();
static void ad_cache_drop() { ad_drop(); }
static void ad_drop()
{
    g_ad.clear();
    if (g_ad_slab) (void)hipFree(g_ad_slab);
    g_ad_slab = nullptr; g_ad_slab_bytes = 0; g_ad_sig.clear();
    if (g_ad_tab) (void)hipFree(g_ad_tab);
    g_ad_tab = nullptr;
}

// dual array of ncomp components over the box of b (16 bytes per entry, the same padding as the library's arrays) out of the
// level's slab, filled with (src, 0) when src is given; the returned pointer is typed double* because BlkView is (the kernels see
// it as Dual*).  Measuring pass: only the size is added up.
static int ad_array(Block* b, AdBlock& a, double** field, const double* src, int ncomp)
{
    const size_t n = (size_t)b->v.nbox * ncomp;
    const size_t bytes = ((n + 32) * 16 + 255) & ~(size_t)255;
    if (g_ad_measure) { g_ad_bump += bytes; *field = nullptr; return 0; }
    if (g_ad_bump + bytes > g_ad_slab_bytes) return fail("forward mode: the slab of dual arrays is too small (internal error)");
    char* raw = g_ad_slab + g_ad_bump;
    g_ad_bump += bytes;
    a.init.push_back(AdInit{raw, src ? src - ADF_PAD0 : nullptr, n, (n + 32) * 16});
    *field = (double*)(raw + (size_t)ADF_PAD0 * 16);
    return 0;
}

// lays the dual arrays of every block of the level out (measuring pass or in the slab)
static int ad_layout(int level, bool viscous)
{
    return for_level(level, [&](Block* b) {
        AdBlock& a = g_ad[b];
        a.v = b->v;
        a.init.clear();
        BlkView& v = a.v;
        const BlkView& p = b->v;
        // active arrays (the seed kernel fills w; the evaluation writes the rest)
        if (ad_array(b, a, &v.w, nullptr, p.nw) || ad_array(b, a, &v.p, nullptr, 1) || ad_array(b, a, &v.rlv, p.rlv, 1) ||
            ad_array(b, a, &v.rev, p.rev, 1) || ad_array(b, a, &v.dw, nullptr, p.nw) || ad_array(b, a, &v.fw, nullptr, 5) ||
            ad_array(b, a, &v.dtl, nullptr, 1) || ad_array(b, a, &v.radI, nullptr, 1) || ad_array(b, a, &v.radJ, nullptr, 1) ||
            ad_array(b, a, &v.radK, nullptr, 1) || ad_array(b, a, &v.ss, p.ss, 1) || ad_array(b, a, &v.scratch, nullptr, 2))
            return 1;
        if (viscous && ad_array(b, a, &v.grad, nullptr, 12)) return 1;
        // passive arrays: value = the library's, derivative 0.  The geometry (x, sI/sJ/sK, vol, volRef, d2wall, dI/dJ/dK) is read by the
        // dual kernels as plain doubles out of the library's own arrays (blkview_def.h: ADF_GEOM), so it has no dual copy
        if (ad_array(b, a, &v.gamma, p.gamma, 1)) return 1;
        v.aa = nullptr; v.wn = v.pn = v.w1 = v.p1 = v.wr = nullptr; v.sFace = nullptr;
        // face arrays of the turbulence boundary treatment
        const size_t nf[3] = {(size_t)p.je * p.ke, (size_t)p.ie * p.ke, (size_t)p.ie * p.je};
        for (int f6 = 0; f6 < 6; ++f6)
            for (int which = 0; which < 2; ++which) {
                double** dst = &(which ? v.bvt : v.bmt)[f6];
                if (!(which ? p.bvt : p.bmt)[f6]) { *dst = nullptr; continue; }
                const size_t bytes = (nf[f6 / 2] * 16 + 255) & ~(size_t)255;
                if (g_ad_measure) { g_ad_bump += bytes; continue; }
                if (g_ad_bump + bytes > g_ad_slab_bytes) return fail("forward mode: the slab of dual arrays is too small (internal error)");
                char* raw = g_ad_slab + g_ad_bump;
                g_ad_bump += bytes;
                a.init.push_back(AdInit{raw, nullptr, nf[f6 / 2], nf[f6 / 2] * 16});
                *dst = (double*)raw;
            }
        return 0;
    });
}

static int ad_prepare(int level)
{
    if (ensure_table(level)) return 1;
    const int maxnn = g_tab_size[level];
    const bool viscous = g_opts.equations != ADFLOW_EULER;
    // the layout the slab must have: level, model, and per block its identity, box, variables and optional arrays
    std::vector<long> sig = {level, viscous ? 1 : 0, maxnn};
    int rc = for_level(level, [&](Block* b) {
        if (!b->geom_uploaded) return fail("geometry of a level-%d block has not been uploaded", level);
        if (viscous && !b->face_vectors_valid) {
            launch_face_vectors(b->v, g_stream);
            b->face_vectors_valid = true;
        }
        const BlkView& p = b->v;
        sig.push_back((long)(uintptr_t)b); sig.push_back(p.nbox); sig.push_back(p.nw);
        sig.push_back((p.d2wall ? 1 : 0) | (p.dI ? 2 : 0) | (p.bmt[0] ? 4 : 0));
        sig.push_back((long)(uintptr_t)p.w); sig.push_back((long)(uintptr_t)p.x);
        return 0;
    });
    if (rc) return rc;
    if (!g_ad_slab || sig != g_ad_sig) {
        ad_drop();
        g_ad_measure = true; g_ad_bump = 0;
        rc = ad_layout(level, viscous);
        g_ad_measure = false;
        if (rc) return rc;
        const size_t need = g_ad_bump + 4096;
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && freeB < need + ((size_t)64 << 20))
            return fail("forward mode needs %.2f GB for the dual copies of the level's arrays, %.2f GB are free on the device", need / 1.e9, freeB / 1.e9);
        HIPCHK(hipMalloc((void**)&g_ad_slab, need));
        g_ad_slab_bytes = need;
        g_ad_bump = 0;
        g_ad.clear();
        if (ad_layout(level, viscous)) { ad_drop(); return 1; }
        std::vector<BlkView> h(maxnn + 1);
        memset(h.data(), 0, sizeof(BlkView) * h.size());
        for (auto& kv : g_blocks)
            if (std::get<0>(kv.first) == level && std::get<1>(kv.first) == 1) h[std::get<2>(kv.first)] = g_ad[kv.second].v;
        HIPCHK(hipMalloc((void**)&g_ad_tab, sizeof(BlkView) * h.size()));
        HIPCHK(hipMemcpyAsync(g_ad_tab, h.data(), sizeof(BlkView) * h.size(), hipMemcpyHostToDevice, g_stream));
        g_ad_sig = sig;
    }
    // values of this call: the passive arrays and the frozen closures from the library's arrays, zero elsewhere
    for (auto& kv : g_ad)
        for (const AdInit& q : kv.second.init) {
            if (q.src) ad_launch_from_real(q.src, q.dst, (long)q.n, g_stream);
            else HIPCHK(hipMemsetAsync(q.dst, 0, q.zeroBytes, g_stream));
        }
    HIPCHK(hipStreamSynchronize(g_stream));
    return 0;
}

// masterRoutines::block_res_state_d (masterRoutines.F90:1285-1393) on the dual arrays of the level: closures with halos,
// turbulence + mean-flow boundary conditions, time step (the spectral radii of the scalar dissipation), SA, fluxes.
// viscPC: block_res_state_d keeps the FULL viscous flux in the preconditioner matrix then (masterRoutines.F90:1380: `.not. lumpedDiss
// .or. viscPC`) -- unlike the finite-difference path block_res_state, whose viscApprox = lumpedDiss whatever viscPC says (:1269-1270)
static int ad_apply_bc_enqueue(int level, const KParams& kp, bool turbBC);
static KParams ad_kparams(int level, unsigned resFlags)
{
    KParams kp = make_kparams(level, 1.0, 0);
    kp.onlyRadii = 1;
    kp.coarseInit = 0;
    kp.dissApprox = (resFlags & ADFLOW_RES_DISS_APPROX) ? 1 : 0;
    return kp;
}
// closuresDone: the seed launch of the caller formed pressure and viscosities already (k_seed_closures)
// the schemes the dual per-face march takes in the exact linearisation (as the plain evaluation chooses, api.hip residual_enqueue:
// scalar JST only with the entropy sensor of NS / RANS on the fine level -- Euler + scalar JST has its own pipelined kernel, which has
// no dual form)
static bool ad_inviscid_march_takes(const KParams& kp)
{
    if (!inviscid_march_enabled() || kp.fwMode) return false;
    if (kp.dissApprox && !kp.fineGrid) return false;
    if (kp.spaceDiscr == ADFLOW_DISS_SCALAR) return inviscid_march_enabled() >= 2 && (kp.viscous || kp.dissApprox) && kp.fineGrid;
    return kp.spaceDiscr == ADFLOW_DISS_MATRIX;       // (upwind: k_roe_march on dual numbers, or the gather kernel)
}

static int ad_block_res_state_enqueue(int level, unsigned resFlags, bool turbBC, bool viscPC, bool closuresDone = false)
{
    KParams kp = ad_kparams(level, resFlags);
    const bool viscApprox = (resFlags & ADFLOW_RES_VISC_APPROX) != 0 && !viscPC;
    LevelTab t;
    if (level_tab(level, &t)) return 1;
    int rc = 0;
    if (!closuresDone) rc = for_level(level, [&](Block* b) {
        ad_launch_closures_halo(g_ad[b].v, kp, g_stream);
        return 0;
    });
    if (rc) return rc;
    if (ad_apply_bc_enqueue(level, kp, turbBC)) return 1;
    // timeStep_block_d: only the scalar dissipation reads the spectral radii; the NS / RANS kernel also leaves the entropy sensor
    // in ss -- not under dissApprox, where ss keeps the frozen sensor of referenceShockSensor (value part, derivative 0)
    if ((resFlags & ADFLOW_RES_FLOW) && kp.spaceDiscr == ADFLOW_DISS_SCALAR)
        ad_launch_time_step_level(g_ad_tab, t.n, t.nx, t.ny, t.nz, kp, g_stream);
    if ((resFlags & ADFLOW_RES_TURB) && g_opts.equations == ADFLOW_RANS) {
        // the marching form on dual numbers (kernels_sa_march.hip; blocks at rest: moving blocks were refused by the caller)
        if (g_sa_march && g_pc_fused) {
            if (ensure_sa_tiles(level)) return 1;
            ad_launch_sa_march(g_ad_tab, g_sa_tiles[level].first, g_sa_tiles[level].second, kp, g_stream);
        } else
            ad_launch_sa_residual_level(g_ad_tab, t.n, t.nx, t.ny, t.nz, kp, g_stream);
    }
    if ((resFlags & ADFLOW_RES_FLOW) && ad_pc_march_applies(kp, viscApprox)) {
        // the preconditioner matrix on the upwind scheme: the dual form of the one-march residual (kernels_pc_march.hip) instead of the
        // gather kernels (blocks at rest: moving blocks were refused by the caller)
        rc = for_level(level, [&](Block* b) {
            if (!b->face_vectors_valid) {
                launch_face_vectors(b->v, g_stream);
                b->face_vectors_valid = true;
            }
            return 0;
        });
        if (rc || ensure_tiles(level)) return 1;
        ad_launch_pc_march(g_ad_tab, g_tiles[level].first, g_tiles[level].second, kp, g_march_kch, g_stream);
    } else if (resFlags & ADFLOW_RES_FLOW) {
        // the Roe upwind scheme of the exact linearisation (second order, the user's limiter): the marching kernel on dual numbers
        // (kernels_roe_march.hip compiled a second time, round 6) instead of the cell-gather kernel -- 3.25 instead of 6 face
        // evaluations and 3.5 instead of 12 reconstructions per cell; it leaves dw + fw in dw for the viscous kernel that follows
        // ... and the full viscous flux as the fused gradient + flux march on dual numbers (k_visc_gf compiled a second time: the metric
        // sums and face geometry stay plain, the ring holds dual gradients -- 160 KB of LDS, one workgroup per CU) instead of the dual
        // gather pair k_nodal_gradients + k_viscous (1.01 ms per pass and 1.3 M cells): in front of the Roe march, which adds its sums
        // (viscFirst), or behind the gather inviscid kernel of the other schemes, completing dw itself
        const bool viscous = kp.viscous && fabs(kp.rFil) >= 1.e-10;
        const bool gfDual = viscous && !viscApprox && g_pc_fused && viscous_is_tiled() >= 2 && kp.fineGrid;
        if (gfDual) {
            rc = for_level(level, [&](Block* b) {
                if (!b->face_vectors_valid) {
                    launch_face_vectors(b->v, g_stream);
                    b->face_vectors_valid = true;
                }
                return 0;
            });
            if (rc || ensure_gf_tiles(level)) return 1;
        }
        bool marched = false;
        if (g_pc_fused && roe_march_takes(kp)) {
            if (ensure_tiles(level)) return 1;
            KParams kr = kp;
            if (gfDual) {
                kr.viscFirst = 1;
                ad_launch_visc_gf(g_ad_tab, g_gf_tiles[level].first, g_gf_tiles[level].second, kr, g_stream);
            }
            marched = ad_launch_roe_march(g_ad_tab, g_tiles[level].first, g_tiles[level].second, kr, g_stream);
            if (marched && gfDual) return 0;
        }
        if (!marched && g_pc_fused && ad_inviscid_march_takes(kp)) {
            // scalar JST with the entropy sensor, matrix dissipation: the per-face march on dual numbers
            // (kernels_inviscid_march.hip compiled a second time) -- four face evaluations per cell instead of the gather kernel's six;
            // behind the viscous march it adds the sums it finds in dw(2:5), as in the plain evaluation
            if (ensure_tiles(level)) return 1;
            KParams ki = kp;
            // the thin-layer viscous flux of the preconditioner matrix marches in front of it as well (k_visc_approx_march on dual
            // numbers), as in the plain approximate residual
            const bool vaDual = viscous && viscApprox && viscous_is_tiled() >= 2 && kp.fineGrid;
            if (vaDual) {
                rc = for_level(level, [&](Block* b) {
                    if (!b->face_vectors_valid) {
                        launch_face_vectors(b->v, g_stream);
                        b->face_vectors_valid = true;
                    }
                    return 0;
                });
                if (rc) return rc;
            }
            if (gfDual) {
                ki.viscFirst = 1;
                ad_launch_visc_gf(g_ad_tab, g_gf_tiles[level].first, g_gf_tiles[level].second, ki, g_stream);
            } else if (vaDual) {
                ki.viscFirst = 1;
                ad_launch_visc_march_approx(g_ad_tab, g_tiles[level].first, g_tiles[level].second, ki, g_stream);
            }
            ad_launch_inviscid_march(g_ad_tab, g_tiles[level].first, g_tiles[level].second, ki, g_stream);
            if (gfDual || vaDual) return 0;
            marched = true;
        }
        if (!marched) ad_launch_inviscid_level(g_ad_tab, t.n, t.nx, t.ny, t.nz, kp, g_stream);
        if (gfDual) {
            ad_launch_visc_gf(g_ad_tab, g_gf_tiles[level].first, g_gf_tiles[level].second, kp, g_stream);
            return 0;
        }
        if (kp.viscous && fabs(kp.rFil) >= 1.e-10) {
            rc = for_level(level, [&](Block* b) {
                if (viscApprox) ad_launch_viscous_approx(g_ad[b].v, kp, g_stream);
                else ad_launch_viscous(g_ad[b].v, kp, g_stream);
                return 0;
            });
            if (rc) return rc;
        }
    }
    return 0;
}

// the device table of the snapshot request: per block slot of the level its snapshot array and its scaled reference residual
static int snap_request_begin(int level, const JacSpec& J)
{
    if (ensure_table(level)) return 1;
    const int n = g_tab_size[level] + 1;
    std::vector<SnapSlot> h((size_t)n, SnapSlot{nullptr});
    for (auto& kv : g_blocks)
        if (std::get<0>(kv.first) == level && std::get<1>(kv.first) == 1) h[std::get<2>(kv.first)] = SnapSlot{kv.second->snap};
    if (g_snapreq.devSlots < n) {
        if (g_snapreq.dev) (void)hipFree(g_snapreq.dev);
        g_snapreq.dev = nullptr; g_snapreq.devSlots = 0;
        HIPCHK(hipMalloc((void**)&g_snapreq.dev, sizeof(SnapSlot) * (size_t)n));
        g_snapreq.devSlots = n;
    }
    HIPCHK(hipStreamSynchronize(g_stream));
    HIPCHK(hipMemcpy(g_snapreq.dev, h.data(), sizeof(SnapSlot) * (size_t)n, hipMemcpyHostToDevice));
    g_snapreq.level = level; g_snapreq.col = 0; g_snapreq.l0 = J.lStart; g_snapreq.n = J.nState;
    g_snapreq.turbScale = g_opts.turbResScale;
    g_snapreq.on = true;
    return 0;
}
struct SnapRequestGuard { ~SnapRequestGuard() { g_snapreq.on = false; } };

int adflow_gpu_fd_jacobian(int level, unsigned flags, double delta)
{
    if (need_ready()) return 1;
    SnapRequestGuard snapGuard;        // no exit leaves the request standing
    if (flags & ~(ADFLOW_JAC_PC | ADFLOW_JAC_FROZEN_TURB | ADFLOW_JAC_TURB_ONLY | ADFLOW_JAC_VISC_PC | ADFLOW_JAC_USE_AD))
        return fail("fd_jacobian: unknown flags 0x%x", flags);
    const bool useAD = (flags & ADFLOW_JAC_USE_AD) != 0;
    if (!useAD && !(delta > 0.0)) return fail("fd_jacobian: delta must be positive");     // forward mode has no step size
    if (useAD) {
        // forward-mode seeds instead of perturbations (adjointUtils.F90:227-409): gather kernels on dual numbers, blocks at rest
        bool moving = false;
        for_level(level, [&](Block* b) { moving = moving || b->v.sFace || b->v.moving; return 0; });
        if (moving) return fail("fd_jacobian(useAD): moving blocks are not linearised (grid velocities)");
        if (!g_act.empty()) return fail("fd_jacobian(useAD): actuator regions are not linearised");
        if (g_bc_callback) return fail("fd_jacobian(useAD): a host boundary-condition hook cannot be linearised");
        if (g_turb_bc_callback && g_opts.equations == ADFLOW_RANS && !(flags & ADFLOW_JAC_FROZEN_TURB))
            return fail("fd_jacobian(useAD): a host turbulence boundary-condition hook cannot be linearised");
    }
    if (level != g_opts.groundLevel) return fail("fd_jacobian: level %d is not the ground level %d (setupStateResidualMatrix sets both)", level, g_opts.groundLevel);
    const bool rans = g_opts.equations == ADFLOW_RANS;
    const bool viscous = rans || g_opts.equations == ADFLOW_NS;
    if ((flags & ADFLOW_JAC_TURB_ONLY) && !rans) return fail("fd_jacobian: ADFLOW_JAC_TURB_ONLY needs the RANS equations");
    if ((flags & ADFLOW_JAC_TURB_ONLY) && (flags & ADFLOW_JAC_FROZEN_TURB)) return fail("fd_jacobian: TURB_ONLY and FROZEN_TURB exclude each other");
    JacSpec J;
    jac_spec(flags & ~ADFLOW_JAC_USE_AD, viscous, rans, &J);
    const int ncomp = J.nStencil * J.nState * J.nState;
    int rc = for_level(level, [&](Block* b) {
        if (!b->wref) {
            if (alloc_arr(b, &b->wref, b->v.nw) || alloc_arr(b, &b->dwref, 6)) return 1;
        }
        if (b->jac_ncomp != ncomp) {
            if (b->jac_raw) HIPCHK(hipFree(b->jac_raw));
            b->jac_raw = nullptr; b->jac = nullptr; b->jac_ncomp = 0;
            const size_t bytes = (size_t)b->v.nbox * ncomp * sizeof(double) + 256;
            HIPCHK(hipMalloc(&b->jac_raw, bytes));
            b->jac = (double*)b->jac_raw + ADF_PAD0;
            b->jac_ncomp = ncomp;
        }
        // (no memset of the blocks: k_fd_scatter stores every entry of every owned row, and only owned rows are ever read)
        // the finite differences of the nColour evaluations of one state variable, scattered into the blocks once per variable
        const int nsnap = J.cn * J.nState;
        if (b->snap_ncomp != nsnap) {
            if (b->snap_raw) HIPCHK(hipFree(b->snap_raw));
            b->snap_raw = nullptr; b->snap = nullptr; b->snap_ncomp = 0;
            HIPCHK(hipMalloc(&b->snap_raw, (size_t)b->v.nbox * nsnap * sizeof(double) + 256));
            b->snap = (double*)b->snap_raw + ADF_PAD0;
            b->snap_ncomp = nsnap;
        }
        return 0;
    });
    if (rc) return rc;
    g_jac_valid = false;

    // whalo2(1, 1, nw, T, T, T) (adjointUtils.F90:113)
    if (g_comm.count(std::make_pair(level, 2)))
        if (halo_exchange_enqueue(level, 1, rans ? 6 : 5, 1, 1, 2)) return 1;

    // the switches of the preconditioner matrix (adjointUtils.F90:176-191), restored below
    const adflow_opts saved = g_opts;
    const int savedLumped = g_lumped;
    unsigned resFlags = 0;
    if (!(flags & ADFLOW_JAC_TURB_ONLY)) resFlags |= ADFLOW_RES_FLOW;
    const bool turbRes = rans && !(flags & ADFLOW_JAC_FROZEN_TURB);
    if (turbRes) resFlags |= ADFLOW_RES_TURB;
    if (flags & ADFLOW_JAC_PC) {
        g_lumped = 1;
        g_opts.acousticScaleFactor = 1.0;
        g_opts.orderTurb = 1;                                           // constants::firstOrder
        resFlags |= ADFLOW_RES_DISS_APPROX | ADFLOW_RES_VISC_APPROX;     // dissApprox = viscApprox = lumpedDiss
        // referenceShockSensor on the unperturbed state (adjointUtils.F90:258-260): the sensor is not linearised
        const bool pressure = (g_opts.equations == ADFLOW_EULER) || (g_opts.spaceDiscr == ADFLOW_DISS_MATRIX);
        rc = for_level(level, [&](Block* b) {
            if (pressure) HIPCHK(hipMemcpyAsync(b->v.ss, b->v.p, sizeof(double) * (size_t)b->boxsize, hipMemcpyDeviceToDevice, g_stream));
            else launch_entropy(b->v, g_stream);
            return 0;
        });
    }
    // frozenTurb: equations = NSEquations (adjointUtils.F90:218-222) -- no turbulence boundary conditions, no SA residual; the eddy
    // viscosity is still recomputed from nuTilde because eddyModel stays set (turbUtils.F90:604-612)
    const bool turbBC = rans && !(flags & ADFLOW_JAC_FROZEN_TURB);
    auto restore = [&]() { g_opts = saved; g_lumped = savedLumped; };
    // Inside the sweep of one state variable over the colours only that component of the state changes; the others are written at its
    // first colour only.  What then differs from a full rewrite is the content of halos BEFORE the boundary conditions of the
    // evaluation overwrite them (the output of the previous evaluation instead of the reference state): face halos are functions of
    // the interior alone, but the halo cells along block EDGES are written by one subface from what another left there, in an order
    // -- so the shortcut is taken for the preconditioner matrix, whose 7-point stencils never reach an edge halo, and not with a host hook
    // (... nor for viscPC, whose full viscous flux reads edge and corner halos through its 27-point stencil: round-5 advisor)
    const bool oneComponent = (flags & ADFLOW_JAC_PC) && !(flags & ADFLOW_JAC_VISC_PC) && !g_bc_callback && !g_turb_bc_callback;

    // the snapshot entries of an evaluation come from the marches only when the host's prediction of the kernel dispatch was right:
    // the launchers report what they wrote (adf_note_snap) and a mismatch is an error instead of a matrix built from stale memory
    auto snap_written = [&](int resFlags_) -> int {
        const int need = ((resFlags_ & ADFLOW_RES_FLOW) ? 1 : 0) | ((resFlags_ & ADFLOW_RES_TURB) ? 2 : 0);
        if ((g_snap_done & need) != need)
            return fail("fd_jacobian: the marching kernels were expected to write the snapshot of the evaluation (need %d, written %d)", need, g_snap_done);
        return 0;
    };
    if (useAD) {
        // one forward-mode evaluation per (colour, state variable): seed = 1 on component l of the cells of the colour (halos
        // included), block_res_state_d, the derivative of the scaled residual is the column of every stencil block
        if (!rc) rc = ad_prepare(level);
        // the marching kernels of the preconditioner matrix write the snapshot of a pass themselves (KParams::snapTab)
        bool snapInMarch = false;
        if (!rc) {
            const KParams kq = ad_kparams(level, resFlags);
            const bool viscApproxA = (resFlags & ADFLOW_RES_VISC_APPROX) != 0 && !(flags & ADFLOW_JAC_VISC_PC);
            snapInMarch = g_jac_snap && (!(resFlags & ADFLOW_RES_FLOW) || ad_pc_march_applies(kq, viscApproxA)) &&
                          (!(resFlags & ADFLOW_RES_TURB) || (g_sa_march && g_pc_fused));
            if (snapInMarch) rc = snap_request_begin(level, J);
        }
        for (int l = J.lStart; l < J.lStart + J.nState && !rc; ++l) {
            for (int col = 0; col < J.cn && !rc; ++col) {
                g_snapreq.col = col;
                const KParams kps = ad_kparams(level, resFlags);
                rc = for_level(level, [&](Block* b) {
                    ad_launch_seed_closures(b->v, g_ad[b].v, l, col, J, kps, g_stream, oneComponent && col > 0);
                    return 0;
                });
                g_snap_done = 0;
                if (!rc) rc = ad_block_res_state_enqueue(level, resFlags, turbBC, (flags & ADFLOW_JAC_VISC_PC) != 0, true);
                if (!rc && snapInMarch) rc = snap_written(resFlags);
                if (!rc && !snapInMarch) rc = for_level(level, [&](Block* b) {
                    ad_launch_snap(b->v, g_ad[b].v.dw, b->snap + (size_t)col * J.nState * b->v.nbox, J, g_opts.turbResScale, g_stream);
                    return 0;
                });
            }
            if (!rc) rc = for_level(level, [&](Block* b) {
                launch_fd_scatter(b->v, b->snap, b->jac, l, J, nullptr, 0.0, g_stream);
                return 0;
            });
        }
        g_snapreq.on = false;
        if (!rc) rc = sync_and_check();
        else (void)hipStreamSynchronize(g_stream);
        if (!g_ad_cache || rc) ad_drop();
        restore();
        if (rc) return rc;
        g_jac = J;
        g_jac_valid = true;
        return 0;
    }

    // setFDReference (adjointUtils.F90:1971-2024): reference residual, then the reference state INCLUDING the halos the boundary
    // conditions just wrote
    bool haveRef = false;           // wref holds the reference state: an error further down still puts the state back
    if (!rc) rc = block_res_state_enqueue(level, resFlags, turbBC);
    if (!rc) rc = for_level(level, [&](Block* b) {
        launch_fd_copy(b->v, b->wref, b->v.w, b->v.nw, g_stream);
        launch_fd_extract(b->v, b->dwref, b->jac, -1, 0, J, 0.0, g_opts.turbResScale, g_stream);
        return 0;
    });
    if (!rc) haveRef = true;
    const double deltaInv = 1.0 / delta;
    // the reference loops colours outside and state variables inside; every (colour, variable) evaluation is independent, so the
    // loops are exchanged here: the nColour evaluations of one variable are kept (dense) and scattered into the blocks together
    const KParams kpc = make_kparams(level, 1.0, 0);
    // the marching kernels of the preconditioner matrix write the snapshot of an evaluation themselves (KParams::snapTab): not with
    // actuator regions (their sources are added to dw behind the core)
    bool snapInMarch = false;
    if (!rc) {
        bool moving = false;
        for_level(level, [&](Block* b) { moving = moving || b->v.sFace || b->v.moving; return 0; });
        snapInMarch = g_jac_snap && g_act.empty() &&
                      (!(resFlags & ADFLOW_RES_FLOW) || pc_march_applies(level, kpc, (resFlags & ADFLOW_RES_VISC_APPROX) != 0)) &&
                      (!(resFlags & ADFLOW_RES_TURB) || (g_sa_march && !moving));
        if (snapInMarch) rc = snap_request_begin(level, J);
    }
    for (int l = J.lStart; l < J.lStart + J.nState && !rc; ++l) {
        for (int col = 0; col < J.cn && !rc; ++col) {
            g_snapreq.col = col;
            rc = for_level(level, [&](Block* b) {
                launch_fd_state_closures(b->v, b->wref, l, col, J, delta, kpc, g_stream, oneComponent && col > 0);
                return 0;
            });
            g_snap_done = 0;
            if (!rc) rc = block_res_state_enqueue(level, resFlags, turbBC, true);
            if (!rc && snapInMarch) rc = snap_written(resFlags);
            if (!rc && !snapInMarch) rc = for_level(level, [&](Block* b) {
                launch_fd_snap(b->v, b->snap + (size_t)col * J.nState * b->v.nbox, J, g_opts.turbResScale, g_stream);
                return 0;
            });
        }
        if (!rc) rc = for_level(level, [&](Block* b) {
            launch_fd_scatter(b->v, b->snap, b->jac, l, J, b->dwref, deltaInv, g_stream);
            return 0;
        });
    }
    g_snapreq.on = false;
    // resetFDReference (adjointUtils.F90:2026-2058): w back, dw = the (scaled) reference residual
    if (!rc) rc = for_level(level, [&](Block* b) {
        launch_fd_state(b->v, b->wref, 0, -1, J, 0.0, g_stream);
        launch_fd_extract(b->v, b->dwref, b->jac, -2, 0, J, 0.0, g_opts.turbResScale, g_stream);
        b->ss_valid = false;
        b->etot_consistent = false;
        return 0;
    });
    if (rc && haveRef) {
        // the sweep stopped half way: the level must not keep a perturbed state (the error text of the failed call is kept)
        for_level(level, [&](Block* b) {
            launch_fd_state(b->v, b->wref, 0, -1, J, 0.0, g_stream);
            b->ss_valid = false;
            b->etot_consistent = false;
            return 0;
        });
        (void)hipStreamSynchronize(g_stream);
    }
    restore();
    if (rc) return rc;
    g_jac = J;
    g_jac_valid = true;
    return sync_and_check();
}

int adflow_gpu_release_workspace(int64_t* bytes)
{
    if (need_ready()) return 1;
    if (g_stream) HIPCHK(hipStreamSynchronize(g_stream));
    if (bytes) *bytes = (int64_t)g_ad_slab_bytes;
    ad_drop();
    return 0;
}

int adflow_gpu_selftest_math(int which, const double* x, const double* a, int64_t n, double* y, double* dy)
{
    if (g_device < 0) return fail("adflow_gpu_init has not been called");
    if (which < 0 || which > 7 || n < 0 || !x || !a || !y || !dy) return fail("selftest_math: bad arguments");
    if (n == 0) return 0;
    double* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, sizeof(double) * 5 * (size_t)n));          // x, a, y, (value, derivative)
    int rc = 0;
    if (hipMemcpy(d, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + n, a, sizeof(double) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess)
        rc = fail("selftest_math: upload failed");
    if (!rc) {
        ad_launch_selftest_math(which, d, d + n, (long)n, d + 2 * n, d + 3 * n, g_stream);
        if (hipStreamSynchronize(g_stream) != hipSuccess || hipMemcpy(y, d + 2 * n, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(dy, d + 3 * n, sizeof(double) * 2 * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail("selftest_math: kernel or download failed");
    }
    (void)hipFree(d);
    return rc;
}

int adflow_gpu_jacobian_info(int32_t* nState, int32_t* nStencil, int32_t* stencil)
{
    if (!g_jac_valid) return fail("jacobian_info: no assembled Jacobian (call adflow_gpu_fd_jacobian first)");
    if (nState) *nState = g_jac.nState;
    if (nStencil) *nStencil = g_jac.nStencil;
    if (stencil)
        for (int s = 0; s < g_jac.nStencil; ++s)
            for (int d = 0; d < 3; ++d) stencil[(size_t)d * g_jac.nStencil + s] = g_jac.st[s][d];
    return 0;
}

int adflow_gpu_download_jacobian(int nn, int level, int sps, double* blocks)
{
    Block* b = find_block(nn, level, sps);
    if (!b) return fail("block (%d,%d,%d) not registered", nn, level, sps);
    if (!g_jac_valid || !b->jac) return fail("download_jacobian: no assembled Jacobian on block (%d,%d,%d)", nn, level, sps);
    if (!blocks) return fail("download_jacobian: blocks is NULL");
    BlkView& v = b->v;
    int rc = copy_box(b, b->jac, blocks, b->jac_ncomp, 2, v.nx, 2, v.ny, 2, v.nz, false);
    if (rc) return rc;
    return sync_and_check();
}

int adflow_gpu_download_jacobian_rows(int nn, int level, int sps, double* rows)
{
    Block* b = find_block(nn, level, sps);
    if (!b) return fail("block (%d,%d,%d) not registered", nn, level, sps);
    if (!g_jac_valid || !b->jac) return fail("download_jacobian_rows: no assembled Jacobian on block (%d,%d,%d)", nn, level, sps);
    if (!rows) return fail("download_jacobian_rows: rows is NULL");
    const BlkView& v = b->v;
    const int nS = g_jac.nState, nSt = g_jac.nStencil;
    if (nS * nS >= 37) return fail("download_jacobian_rows: nState = %d exceeds the tile of the transposition kernel", nS);
    const size_t perPlane = (size_t)v.nx * v.ny * nSt * nS * nS;          // doubles of one k plane of rows
    int nk = (int)std::max<size_t>(1, ((size_t)128 << 20) / (perPlane * 8));   // slabs of <= 128 MB on the device
    if (nk > v.nz) nk = v.nz;
    double* d_out = nullptr;
    HIPCHK(hipMalloc((void**)&d_out, perPlane * nk * 8));
    for (int k0 = 0; k0 < v.nz; k0 += nk) {
        const int n = std::min(nk, v.nz - k0);
        launch_jac_rows(v, b->jac, d_out, nS, nSt, k0 + 2, n, g_stream);
        if (hipMemcpyAsync(rows + perPlane * k0, d_out, perPlane * n * 8, hipMemcpyDeviceToHost, g_stream) != hipSuccess ||
            hipStreamSynchronize(g_stream) != hipSuccess) {
            (void)hipFree(d_out);
            return fail("download_jacobian_rows: copy of the planes %d.. failed", k0 + 2);
        }
    }
    HIPCHK(hipFree(d_out));
    return sync_and_check();
}

}  // extern "C"

// ------------------------------------------------------------ halo exchange
namespace {

struct LevelDims { int nx, ny, nz; };
std::map<int, LevelDims> g_tab_dims;                   // largest block extents of a level (grid of the batched launches)

int ensure_table(int level)
{
    if (g_tab.count(level)) return 0;
    int maxnn = 0;
    for (auto& kv : g_blocks)
        if (std::get<0>(kv.first) == level) maxnn = std::max(maxnn, std::get<2>(kv.first));
    if (maxnn == 0) return fail("no block registered on level %d", level);
    std::vector<BlkView> h(maxnn + 1);
    memset(h.data(), 0, sizeof(BlkView) * h.size());
    for (auto& kv : g_blocks)
        if (std::get<0>(kv.first) == level && std::get<1>(kv.first) == 1) h[std::get<2>(kv.first)] = kv.second->v;
    long off = 0;                     // PETSc vector order: block nn ascending (NKSolvers.F90:1240-1253)
    for (auto& v : h) {
        v.vecOff = off;
        off += (long)v.nx * v.ny * v.nz * v.nw;
    }
    BlkView* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, sizeof(BlkView) * h.size()));
    HIPCHK(hipMemcpy(d, h.data(), sizeof(BlkView) * h.size(), hipMemcpyHostToDevice));
    g_tab[level] = d;
    g_tab_size[level] = maxnn;
    LevelDims ld = {0, 0, 0};
    for (auto& v : h) { ld.nx = std::max(ld.nx, v.nx); ld.ny = std::max(ld.ny, v.ny); ld.nz = std::max(ld.nz, v.nz); }
    g_tab_dims[level] = ld;
    return 0;
}

// level-batched pointwise launches: device block table + the largest block extents of the level
int level_tab(int level, LevelTab* t)
{
    if (ensure_table(level)) return 1;
    const LevelDims& ld = g_tab_dims[level];
    t->tab = g_tab[level]; t->n = g_tab_size[level]; t->nx = ld.nx; t->ny = ld.ny; t->nz = ld.nz;
    return 0;
}

int time_step_level(int level, const KParams& kp)
{
    LevelTab t;
    if (level_tab(level, &t)) return 1;
    launch_time_step_level(t.tab, t.n, t.nx, t.ny, t.nz, kp, g_stream);
    // the NS / RANS kernel also leaves the entropy sensor in ss -- but not in a dissApprox pass (frozen sensor kept)
    const bool wrote_ss = kp.viscous && !kp.dissApprox;
    for (auto& kv : g_blocks)
        if (std::get<0>(kv.first) == level) kv.second->ss_valid = wrote_ss;
    return 0;
}

// implicit residual averaging of every block of the level (one launch per direction)
// scaleDtl != 0: dw enters as scaleDtl dtl dw (the stage scaling of the Runge-Kutta smoother): folded into the first sweep when every
// block of the level has an i sweep, a pointwise pass in front otherwise
int res_averaging_level(int level, const KParams& kp, double scaleDtl)
{
    if (ensure_table(level)) return 1;
    const LevelDims& ld = g_tab_dims[level];
    bool fold = (scaleDtl != 0.0);
    for_level(level, [&](Block* b) { fold = fold && b->v.nx > 1; return 0; });
    if (scaleDtl != 0.0 && !fold) launch_scale_dw_level(g_tab[level], g_tab_size[level], ld.nx, ld.ny, ld.nz, scaleDtl, 0, g_stream);
    launch_res_averaging_level(g_tab[level], g_tab_size[level], ld.nx, ld.ny, ld.nz, kp, g_stream, fold ? scaleDtl : 0.0);
    return 0;
}

// tile table of the k-marching Euler kernel for every block of a level.  The
// dispatcher places workgroup p on XCD p % 8 (MI355X_MICROARCH.md): entry p holds
// the tile  (p % 8) * ceil(n/8) + p / 8  of the natural order (i-tile fastest, then
// j, k, block) so that each XCD owns one contiguous slab and j/k-neighbouring tiles
// share its 4 MiB L2.
}  // namespace

// workgroups of a marching kernel resident at a time: two per CU (256 VGPRs, <= 80 KB of LDS each)
void adf_note_rvec(int bits) { g_rvec_done |= bits; }
void adf_note_snap(int bits) { g_snap_done |= bits; }

int adf_round_size()
{
    if (g_num_cus <= 0) {
        hipDeviceProp_t prop;
        if (g_device >= 0 && hipGetDeviceProperties(&prop, g_device) == hipSuccess) g_num_cus = prop.multiProcessorCount;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    return 2 * g_num_cus;
}

namespace {

// Launch order of T work items that run in rounds of W resident workgroups: slot p of the table -> item (or -1 = empty slot).  Every
// round occupies Wp = W rounded up to a multiple of 8 slots, so that slot p is dispatched to XCD p % 8 in every round whatever the CU
// count (round-3 advisor finding: with W % 8 != 0 the plain (p % W) / 8, (p % W) % 8 decode dropped items); inside a round XCD x takes
// the x-th contiguous eighth of the round's items.  Checked: every item exactly once.
int round_order(int T, int W, std::vector<int>* out)
{
    if (W < 1) W = 1;
    const int Wp = (W + 7) / 8 * 8, rounds = (T + W - 1) / W;
    out->assign((size_t)rounds * Wp, -1);
    long seen = 0;
    for (int q = 0; q < rounds; ++q) {
        const int nIn = std::min(W, T - q * W), per = (nIn + 7) / 8;
        for (int pr = 0; pr < Wp; ++pr) {
            const int s = pr / 8, x = pr % 8;
            if (s < per && x * per + s < nIn) { (*out)[(size_t)q * Wp + pr] = q * W + x * per + s; ++seen; }
        }
    }
    if (seen != T) return fail("round_order: %ld of %d work items placed (W = %d)", seen, T, W);
    std::vector<char> hit((size_t)T, 0);
    for (int e : *out)
        if (e >= 0) { if (hit[e]) return fail("round_order: item %d placed twice", e); hit[e] = 1; }
    return 0;
}

int ensure_tiles(int level)
{
    if (g_tiles.count(level)) return 0;
    if (ensure_table(level)) return 1;
    std::vector<int4> nat;
    for (auto& kv : g_blocks) {
        if (std::get<0>(kv.first) != level || std::get<1>(kv.first) != 1) continue;
        int ntx, nty, ntz;
        euler_march_tiles(kv.second->v, &ntx, &nty, &ntz);
        for (int tz = 0; tz < ntz; ++tz)
            for (int ty = 0; ty < nty; ++ty)
                for (int tx = 0; tx < ntx; ++tx) {
                    int4 t;
                    t.x = std::get<2>(kv.first); t.y = tx; t.z = ty; t.w = tz;
                    nat.push_back(t);
                }
    }
    // workgroup p runs on XCD p % 8.  xcd_tiles = 1: XCD x takes the x-th contiguous eighth of the whole table; 2 (default): of every
    // ROUND of resident workgroups (adf_round_size), so that at any time the eight XCDs work on neighbouring tiles of one block
    // (no measurable difference to 1 on the north-star mesh, profiles/r03_d_xcd_round_ab.txt; it keeps the working set of a round
    // compact when a level holds many blocks)
    const int n = (int)nat.size();
    const int W = (g_xcd_tiles >= 2) ? adf_round_size() : ((n + 7) / 8) * 8;
    std::vector<int> ord;
    if (g_xcd_tiles) {
        if (round_order(n, W, &ord)) return 1;
    } else {
        ord.resize(n);
        for (int p = 0; p < n; ++p) ord[p] = p;
    }
    std::vector<int4> phys(ord.size());
    for (size_t p = 0; p < ord.size(); ++p) {
        if (ord[p] >= 0) phys[p] = nat[ord[p]];
        else { phys[p].x = -1; phys[p].y = phys[p].z = phys[p].w = 0; }
    }
    while (!phys.empty() && phys.back().x < 0) phys.pop_back();
    int4* d = nullptr;
    HIPCHK(hipMalloc((void**)&d, sizeof(int4) * phys.size()));
    HIPCHK(hipMemcpy(d, phys.data(), sizeof(int4) * phys.size(), hipMemcpyHostToDevice));
    g_tiles[level] = std::make_pair(d, (int)phys.size());
    return 0;
}

// Chunk table of the fused gradient + viscous march (k_visc_gf).  A column = (block, 60-column tile, 3-row tile); every column is cut
// into chunks of k planes, one workgroup per chunk, and a workgroup costs (planes + ~1.5 warm-up planes).  The device keeps
// W = 2 x CUs workgroups of this kernel resident and dispatches them in launch order, so the launch runs in "rounds" of W workgroups
// of about equal duration: with N columns x c chunks just above a multiple of W the last round is almost empty (north-star mesh:
// 1032 columns x 2 chunks = 4.03 rounds = the time of 5).  Here the NUMBER of chunks is chosen as a whole number of rounds, columns
// get c or c+1 chunks, the chunks are launched longest first (every round holds chunks of one length), and within a round
// workgroup p = 8 s + x takes the x-th contiguous eighth of the round's chunks, so that XCD x (which receives the workgroups
// p % 8 == x) works on neighbouring tiles.  Entry: x = block slot (-1: empty), y = bx | by << 16, z = first plane, w = last plane.
// R = produced cell rows per workgroup (3: k_visc_gf, 4: k_sa_march), warm = planes a chunk costs beyond its own, reach = cells the
// stencil of a produced cell reaches (the interior / boundary partition); all / in / bd: the three tables
static int build_chunk_tables(int level, int R, double warm, int reach, std::pair<int4*, int>* all, std::pair<int4*, int>* in_,
                              std::pair<int4*, int>* bd_, int wgPerCU = 2)
{
    if (ensure_table(level)) return 1;
    struct Col { int slot, bx, by, nz, nx, ny; };
    std::vector<Col> cols;
    long planes = 0;
    for (auto& kv : g_blocks) {
        if (std::get<0>(kv.first) != level || std::get<1>(kv.first) != 1) continue;
        const BlkView& v = kv.second->v;
        const int gx = (v.nx + 59) / 60, gy = (v.ny + R - 1) / R;
        for (int by = 0; by < gy; ++by)
            for (int bx = 0; bx < gx; ++bx) { cols.push_back(Col{std::get<2>(kv.first), bx, by, v.nz, v.nx, v.ny}); planes += v.nz; }
    }
    const int N = (int)cols.size();
    *all = *in_ = *bd_ = std::make_pair((int4*)nullptr, 0);
    if (N == 0) return 0;
    const int W = adf_round_size() * wgPerCU / 2;       // resident workgroups of the kernel: wgPerCU per CU
    const int kmin = 8;
    // chunks of a column for a total of T chunks: proportional to its planes (largest remainder), at least 1, at most nz / kmin
    auto split = [&](long T, std::vector<int>& c) {
        c.assign(N, 1);
        long used = 0;
        std::vector<std::pair<double, int>> rem(N);
        for (int q = 0; q < N; ++q) {
            const double want = (double)T * cols[q].nz / (double)planes;
            const int cmax = std::max(1, cols[q].nz / kmin);
            c[q] = std::min(cmax, std::max(1, (int)want));
            rem[q] = std::make_pair(want - c[q], q);
            used += c[q];
        }
        std::sort(rem.begin(), rem.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first > b.first; });
        for (int q = 0; q < N && used < T; ++q) {
            const int id = rem[q].second;
            if (c[id] < std::max(1, cols[id].nz / kmin)) { ++c[id]; ++used; }
        }
    };
    auto makespan = [&](const std::vector<int>& c) {
        std::vector<int> len;
        for (int q = 0; q < N; ++q)
            for (int e = 0; e < c[q]; ++e) len.push_back((cols[q].nz + c[q] - 1 - e) / c[q]);
        std::sort(len.begin(), len.end(), [](int a, int b) { return a > b; });
        double t = 0.0;
        for (size_t q = 0; q < len.size(); q += W) t += len[q] + warm;
        return t;
    };
    std::vector<int> best;
    if (!g_gf_nofit) {
        double tbest = 1.e300;
        const long Tmax = std::max<long>(N, planes / kmin);
        for (long m = 1; m * W <= Tmax + W; ++m) {
            std::vector<int> c;
            split(m * W, c);
            const double t = makespan(c);
            if (t < tbest - 1.e-9) { tbest = t; best = c; }
            if (m > 64) break;
        }
        // a level much smaller than the device: one chunk per column unless splitting fills more of it
        std::vector<int> one(N, 1);
        if (makespan(one) < tbest) best = one;
    } else {
        best.resize(N);
        const int L = g_march_kch > 0 ? g_march_kch : 32;
        for (int q = 0; q < N; ++q) best[q] = (cols[q].nz + L - 1) / L;
    }
    struct Chunk { int col, k0, k1; };
    std::vector<Chunk> ch;
    for (int q = 0; q < N; ++q) {
        int k = 2;
        for (int e = 0; e < best[q]; ++e) {
            const int len = (cols[q].nz + best[q] - 1 - e) / best[q];
            if (len <= 0) continue;
            ch.push_back(Chunk{q, k, k + len - 1});
            k += len;
        }
    }
    // longest first; chunks of one length in (block, k, row tile, column tile) order
    std::stable_sort(ch.begin(), ch.end(), [&](const Chunk& a, const Chunk& b) {
        const int la = a.k1 - a.k0, lb = b.k1 - b.k0;
        if (la != lb) return la > lb;
        if (cols[a.col].slot != cols[b.col].slot) return cols[a.col].slot < cols[b.col].slot;
        if (a.k0 != b.k0) return a.k0 < b.k0;
        return a.col < b.col;
    });
    // round-ordered device table of a chunk list
    auto make_table = [&](const std::vector<Chunk>& list, std::pair<int4*, int>* out) -> int {
        const int T = (int)list.size();
        std::vector<int> ord;
        if (round_order(T, W, &ord)) return 1;
        std::vector<int4> phys(ord.size());
        for (size_t p = 0; p < ord.size(); ++p) {
            const int e = ord[p];
            int4 tt;
            if (e >= 0) {
                const Col& c = cols[list[e].col];
                tt.x = c.slot; tt.y = c.bx | (c.by << 16); tt.z = list[e].k0; tt.w = list[e].k1;
            } else { tt.x = -1; tt.y = tt.z = tt.w = 0; }
            phys[p] = tt;
        }
        int n = (int)phys.size();                              // trailing empty entries are not launched
        while (n > 0 && phys[n - 1].x < 0) --n;
        int4* d = nullptr;
        HIPCHK(hipMalloc((void**)&d, sizeof(int4) * std::max(n, 1)));
        if (n > 0) HIPCHK(hipMemcpy(d, phys.data(), sizeof(int4) * n, hipMemcpyHostToDevice));
        *out = std::make_pair(d, n);
        return 0;
    };
    if (make_table(ch, all)) return 1;
    // the same chunks in two tables for the evaluation split around the halo exchange: "interior" = the produced cells (columns
    // 2+60 bx .., rows 2+R by .., planes k0 .. k1) and their stencil (+-reach) lie inside the owned range
    std::vector<Chunk> in, bd;
    for (const Chunk& c : ch) {
        const Col& q = cols[c.col];
        const int il = q.nx + 1, jl = q.ny + 1, kl = q.nz + 1;
        const int ia = 2 + 60 * q.bx, ib_ = std::min(ia + 59, il), ja = 2 + R * q.by, jb_ = std::min(ja + R - 1, jl);
        const bool interior = (ia - reach >= 2 && ib_ + reach <= il && ja - reach >= 2 && jb_ + reach <= jl && c.k0 - reach >= 2 &&
                               c.k1 + reach <= kl);
        (interior ? in : bd).push_back(c);
    }
    if (make_table(in, in_) || make_table(bd, bd_)) return 1;
    return 0;
}

int ensure_gf_tiles(int level)
{
    if (g_gf_tiles.count(level)) return 0;
    std::pair<int4*, int> a, i, b;
    if (build_chunk_tables(level, 3, 1.5, 1, &a, &i, &b, 2)) return 1;
    g_gf_tiles[level] = a; g_gf_tiles_int[level] = i; g_gf_tiles_bnd[level] = b;
    return 0;
}

// chunk tables of the Spalart-Allmaras march: 60 columns x 4 rows, no warm-up plane beyond the window fill, stencil +-2
int ensure_sa_tiles(int level)
{
    if (g_sa_tiles.count(level)) return 0;
    std::pair<int4*, int> a, i, b;
    if (build_chunk_tables(level, 4, 0.5, 2, &a, &i, &b)) return 1;
    g_sa_tiles[level] = a; g_sa_tiles_int[level] = i; g_sa_tiles_bnd[level] = b;
    return 0;
}

int make_list_host(int level, const int32_t* blk, const int32_t* idx, int ld, int first, int n, std::vector<int>& hb,
                   std::vector<long>& ho)
{
    hb.resize(n);
    ho.resize(n);
    for (int t = 0; t < n; ++t) {
        const int nn = blk[first + t];
        Block* b = find_block(nn, level, 1);
        if (!b) return fail("comm pattern of level %d references unregistered block %d", level, nn);
        const int i = idx[first + t], j = idx[ld + first + t], k = idx[2 * ld + first + t];
        if (i < 0 || i > b->v.ib || j < 0 || j > b->v.jb || k < 0 || k > b->v.kb)
            return fail("comm pattern of level %d: cell (%d,%d,%d) outside block %d", level, i, j, k, nn);
        hb[t] = nn;
        ho[t] = b->v.idx(i, j, k);
    }
    return 0;
}

int upload_list(const std::vector<int>& hb, const std::vector<long>& ho, int** d_blk, long** d_off)
{
    const size_t n = hb.size();
    *d_blk = nullptr;
    *d_off = nullptr;
    if (n == 0) return 0;
    HIPCHK(hipMalloc((void**)d_blk, sizeof(int) * n));
    HIPCHK(hipMalloc((void**)d_off, sizeof(long) * n));
    HIPCHK(hipMemcpy(*d_blk, hb.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(*d_off, ho.data(), sizeof(long) * n, hipMemcpyHostToDevice));
    return 0;
}

int make_list(int level, const int32_t* blk, const int32_t* idx, int ld, int first, int n, int** d_blk, long** d_off)
{
    std::vector<int> hb;
    std::vector<long> ho;
    if (make_list_host(level, blk, idx, ld, first, n, hb, ho)) return 1;
    return upload_list(hb, ho, d_blk, d_off);
}

int build_comm(int level, int nLayers, CommPattern** out)
{
    auto it = g_comm.find(std::make_pair(level, nLayers));
    if (it == g_comm.end()) return fail("no %d-layer comm pattern registered for level %d", nLayers, level);
    CommPattern& cp = it->second;
    *out = &cp;
    if (ensure_table(level)) return 1;
    if (cp.built) return 0;
    const int nc = (int)cp.h_donorBlock.size();
    cp.local.n = nc;
    {
        // same-GPU copies: the (donor, halo) pairs are independent (donors are owned cells), so they are
        // re-ordered by halo address: consecutive lanes then write - and for 1-to-1 matching blocks also
        // read - consecutive memory whatever order the host's commPattern lists them in
        std::vector<int> db, hb;
        std::vector<long> dof, hof;
        if (make_list_host(level, cp.h_donorBlock.data(), cp.h_donorIdx.data(), nc, 0, nc, db, dof)) return 1;
        if (make_list_host(level, cp.h_haloBlock.data(), cp.h_haloIdx.data(), nc, 0, nc, hb, hof)) return 1;
        std::vector<int> perm(nc);
        for (int t = 0; t < nc; ++t) perm[t] = t;
        std::sort(perm.begin(), perm.end(), [&](int a, int c) { return hb[a] != hb[c] ? hb[a] < hb[c] : hof[a] < hof[c]; });
        std::vector<int> db2(nc), hb2(nc);
        std::vector<long> dof2(nc), hof2(nc);
        for (int t = 0; t < nc; ++t) { db2[t] = db[perm[t]]; dof2[t] = dof[perm[t]]; hb2[t] = hb[perm[t]]; hof2[t] = hof[perm[t]]; }
        if (g_comm_self && nc > 0) {
            // tuning "comm_self": the same-process interfaces take the inter-GPU path -- k_halo_pack, ncclSend / ncclRecv to the own
            // rank inside the group, k_halo_unpack -- so that the RCCL leg of whalo1 / whalo2 executes on a single GPU
            // (the t-th packed donor is the t-th unpacked halo)
            cp.local.n = 0;
            CommList s, r;
            s.n = r.n = nc;
            s.peer = r.peer = g_self_rank;
            if (upload_list(db2, dof2, &s.blkA, &s.offA)) return 1;
            if (upload_list(hb2, hof2, &r.blkA, &r.offA)) return 1;
            HIPCHK(hipMalloc((void**)&s.buf, sizeof(double) * 11 * (size_t)nc));
            HIPCHK(hipMalloc((void**)&r.buf, sizeof(double) * 11 * (size_t)nc));
            cp.sends.push_back(s);
            cp.recvs.push_back(r);
        } else {
            if (upload_list(db2, dof2, &cp.local.blkA, &cp.local.offA)) return 1;
            if (upload_list(hb2, hof2, &cp.local.blkB, &cp.local.offB)) return 1;
        }
    }
    const int ns = (int)cp.h_sendProc.size(), nr = (int)cp.h_recvProc.size();
    const int nst = ns ? cp.h_nsendCum[ns] : 0, nrt = nr ? cp.h_nrecvCum[nr] : 0;
    const int s0 = (int)cp.sends.size(), r0 = (int)cp.recvs.size();       // the self message, when there is one
    cp.sends.resize(s0 + ns);
    cp.recvs.resize(r0 + nr);
    for (int i = 0; i < ns; ++i) {
        CommList& l = cp.sends[s0 + i];
        l.peer = cp.h_sendProc[i];
        l.n = cp.h_nsendCum[i + 1] - cp.h_nsendCum[i];
        if (make_list(level, cp.h_sendBlock.data(), cp.h_sendIdx.data(), nst, cp.h_nsendCum[i], l.n, &l.blkA, &l.offA)) return 1;
        HIPCHK(hipMalloc((void**)&l.buf, sizeof(double) * 11 * (size_t)std::max(l.n, 1)));
    }
    for (int i = 0; i < nr; ++i) {
        CommList& l = cp.recvs[r0 + i];
        l.peer = cp.h_recvProc[i];
        l.n = cp.h_nrecvCum[i + 1] - cp.h_nrecvCum[i];
        if (make_list(level, cp.h_recvBlock.data(), cp.h_recvIdx.data(), nrt, cp.h_nrecvCum[i], l.n, &l.blkA, &l.offA)) return 1;
        HIPCHK(hipMalloc((void**)&l.buf, sizeof(double) * 11 * (size_t)std::max(l.n, 1)));
    }
    for (auto& pd : cp.periodic) {
        free_list(pd.list);
        pd.list.n = (int)pd.h_block.size();
        if (pd.list.n > 0 && make_list(level, pd.h_block.data(), pd.h_idx.data(), pd.list.n, 0, pd.list.n, &pd.list.blkA, &pd.list.offA))
            return 1;
    }
    cp.built = true;
    return 0;
}

// setCommPointers (haloExchange.F90:392-415): which variables travel
int halo_mask(int varStart, int varEnd, int commPressure, int commVisc, unsigned* mask, int* nvar)
{
    if (varStart < 1 || varEnd > 6) return fail("halo exchange: variable range %d..%d outside 1..6", varStart, varEnd);
    unsigned m = 0;
    int n = 0;
    for (int l = varStart; l <= varEnd; ++l) { m |= 1u << (l - 1); ++n; }
    if (commPressure) { m |= 1u << 8; ++n; }
    const bool visc = (g_opts.equations == ADFLOW_NS || g_opts.equations == ADFLOW_RANS);
    if (visc && commVisc) { m |= 1u << 9; ++n; }
    if (g_opts.equations == ADFLOW_RANS && commVisc) { m |= 1u << 10; ++n; }
    *mask = m;
    *nvar = n;
    return 0;
}

}  // namespace

#ifndef ADFLOW_NO_RCCL
#include <rccl/rccl.h>
namespace {
ncclComm_t g_nccl = nullptr;
int g_rank = 0, g_nranks = 1;
}
#define NCCLCHK(expr)                                                                          \
    do {                                                                                       \
        ncclResult_t r_ = (expr);                                                              \
        if (r_ != ncclSuccess) return fail("%s failed: %s", #expr, ncclGetErrorString(r_));    \
    } while (0)
#endif

extern "C" {

int adflow_gpu_set_bc_callback(adflow_bc_callback fn)
{
    ++g_state_gen;
    g_bc_callback = fn;
    return 0;
}

int adflow_gpu_bc_register(int nn, int level, int sps, int nBocos, int nViscBocos, const adflow_bc_subface* faces)
{
    ++g_state_gen;
    Block* b = find_block(nn, level, sps);
    if (!b) return fail("block (%d,%d,%d) not registered", nn, level, sps);
    if (nBocos < 0 || nViscBocos < 0 || nViscBocos > nBocos) return fail("bc_register: nBocos=%d nViscBocos=%d", nBocos, nViscBocos);
    if (nBocos > 0 && !faces) return fail("bc_register: null subface list");
    const BlkView& v = b->v;
    std::vector<BcFaceDev> out;
    std::vector<void*> fresh;       // device arrays of this registration
    auto drop_fresh = [&]() { for (void* q : fresh) (void)hipFree(q); };
    for (int m = 0; m < nBocos; ++m) {
        const adflow_bc_subface& f = faces[m];
        switch (f.bcType) {
        case ADFLOW_BC_SYMM: case ADFLOW_BC_NSWALL_ADIABATIC: case ADFLOW_BC_NSWALL_ISOTHERMAL: case ADFLOW_BC_EULERWALL:
        case ADFLOW_BC_FARFIELD: case ADFLOW_BC_SUPERSONIC_INFLOW: case ADFLOW_BC_SUPERSONIC_OUTFLOW: case ADFLOW_BC_EXTRAP:
        case ADFLOW_BC_SYMM_POLAR: case ADFLOW_BC_SUBSONIC_INFLOW: case ADFLOW_BC_SUBSONIC_OUTFLOW: case ADFLOW_BC_MASSBLEED_OUTFLOW:
            break;
        default:
            return drop_fresh(), fail("bc_register: block %d subface %d: BCType %d is not implemented on the device "
                        "(inflow bleeds, mDot / thrust, domain and sliding interfaces stay with the host callback)", nn, m + 1, f.bcType);
        }
        if (f.faceID < ADFLOW_IMIN || f.faceID > ADFLOW_KMAX) return drop_fresh(), fail("bc_register: block %d subface %d: BCFaceID %d", nn, m + 1, f.faceID);
        // generic subface indices run over the two in-plane directions of the block face (utils.F90:881-1175)
        const int amax = (f.faceID <= ADFLOW_IMAX) ? v.jb : v.ib;
        const int bmax = (f.faceID <= ADFLOW_JMAX) ? v.kb : v.jb;
        if (f.icBeg < 0 || f.icEnd > amax || f.jcBeg < 0 || f.jcEnd > bmax || f.icEnd < f.icBeg || f.jcEnd < f.jcBeg)
            return drop_fresh(), fail("bc_register: block %d subface %d: cell range %d:%d x %d:%d outside the block face", nn, m + 1, f.icBeg,
                        f.icEnd, f.jcBeg, f.jcEnd);
        const bool subOut = (f.bcType == ADFLOW_BC_SUBSONIC_OUTFLOW || f.bcType == ADFLOW_BC_MASSBLEED_OUTFLOW);
        const bool subIn = (f.bcType == ADFLOW_BC_SUBSONIC_INFLOW);
        const bool needNorm = (f.bcType == ADFLOW_BC_SYMM || f.bcType == ADFLOW_BC_EULERWALL || f.bcType == ADFLOW_BC_FARFIELD || subOut || subIn);
        if (subOut && !f.ps) return drop_fresh(), fail("bc_register: block %d subface %d: BCData%%ps is required", nn, m + 1);
        if (subIn) {
            if (f.subsonicInletTreatment == ADFLOW_INLET_TOTAL_CONDITIONS) {
                if (!(f.ptInlet && f.ttInlet && f.htInlet && f.flowXdirInlet && f.flowYdirInlet && f.flowZdirInlet))
                    return drop_fresh(), fail("bc_register: block %d subface %d: ptInlet, ttInlet, htInlet, flow[XYZ]dirInlet are required", nn, m + 1);
            } else if (f.subsonicInletTreatment == ADFLOW_INLET_MASS_FLOW) {
                if (!(f.rho && f.velx && f.vely && f.velz)) return drop_fresh(), fail("bc_register: block %d subface %d: rho, velx, vely, velz are required", nn, m + 1);
            } else
                return drop_fresh(), fail("bc_register: block %d subface %d: subsonicInletTreatment=%d (1 total conditions, 2 mass flow)", nn, m + 1,
                            f.subsonicInletTreatment);
        }
        if (f.bcType == ADFLOW_BC_SYMM_POLAR && !v.x) return drop_fresh(), fail("bc_register: block %d subface %d: symmPolar needs the node coordinates x", nn, m + 1);
        if ((subIn || f.bcType == ADFLOW_BC_SUPERSONIC_INFLOW) && v.nw > 5 && !f.turbInlet)
            return drop_fresh(), fail("bc_register: block %d subface %d: BCData%%turbInlet is required for an inflow subface of a RANS block", nn, m + 1);
        if (needNorm && !f.norm) return drop_fresh(), fail("bc_register: block %d subface %d: BCData%%norm is required", nn, m + 1);
        if (f.bcType == ADFLOW_BC_NSWALL_ISOTHERMAL && !f.TNS_Wall) return drop_fresh(), fail("bc_register: block %d subface %d: TNS_Wall is required", nn, m + 1);
        if (f.bcType == ADFLOW_BC_SUPERSONIC_INFLOW && !(f.rho && f.velx && f.vely && f.velz && f.ps))
            return drop_fresh(), fail("bc_register: block %d subface %d: rho, velx, vely, velz, ps are required", nn, m + 1);
        const size_t n = (size_t)(f.icEnd - f.icBeg + 1) * (f.jcEnd - f.jcBeg + 1);
        auto up = [&](const double* h, int nc, const double** dev) -> int {
            *dev = nullptr;
            if (!h) return 0;
            void* raw = nullptr;
            HIPCHK(hipMalloc(&raw, sizeof(double) * n * nc));
            HIPCHK(hipMemcpy(raw, h, sizeof(double) * n * nc, hipMemcpyHostToDevice));
            fresh.push_back(raw);
            *dev = (const double*)raw;
            return 0;
        };
        BcFaceDev d;
        d.type = f.bcType; d.faceID = f.faceID; d.icBeg = f.icBeg; d.icEnd = f.icEnd; d.jcBeg = f.jcBeg; d.jcEnd = f.jcEnd;
        if (up(f.norm, 3, &d.norm) || up(f.rface, 1, &d.rface) || up(f.uSlip, 3, &d.uslip) || up(f.TNS_Wall, 1, &d.tns) ||
            up(f.rho, 1, &d.rho) || up(f.velx, 1, &d.vx) || up(f.vely, 1, &d.vy) || up(f.velz, 1, &d.vz) || up(f.ps, 1, &d.ps) ||
            up(f.ptInlet, 1, &d.pt) || up(f.ttInlet, 1, &d.tt) || up(f.htInlet, 1, &d.ht) || up(f.flowXdirInlet, 1, &d.fdx) ||
            up(f.flowYdirInlet, 1, &d.fdy) || up(f.flowZdirInlet, 1, &d.fdz) || up(v.nw > 5 ? f.turbInlet : nullptr, 1, &d.turbInlet))
            return 1;
        d.inletTreatment = f.subsonicInletTreatment; d.pad = 0;
        for (int q = 0; q < 3; ++q) d.symNorm[q] = f.symNorm[q];
        d.tauq = nullptr;
        if (m < nViscBocos) {
            int r[4];
            bc_owned_range(f.faceID, f.icBeg, f.icEnd, f.jcBeg, f.jcEnd, v.il, v.jl, v.kl, r);
            const long no = (long)std::max(0, r[1] - r[0] + 1) * std::max(0, r[3] - r[2] + 1);
            if (no > 0) {
                void* raw = nullptr;
                HIPCHK(hipMalloc(&raw, sizeof(double) * 9 * no));
                HIPCHK(hipMemsetAsync(raw, 0, sizeof(double) * 9 * no, g_stream));
                fresh.push_back(raw);
                d.tauq = (double*)raw;
            }
        }
        out.push_back(d);
    }
    // the previous registration's device data is no longer referenced once the plan is dropped
    bc_plan_drop(level);
    if (g_stream) (void)hipStreamSynchronize(g_stream);
    for (void* q : b->bc_allocs) (void)hipFree(q);
    b->bc_allocs = fresh;
    b->bc = out;
    b->nViscBocos = nViscBocos;
    if (v.nw > 5 && !v.bmt[0]) {
        // face arrays of the implicit turbulence boundary treatment (bmt/bvt of blockPointers, one turbulence variable)
        const size_t nf[3] = {(size_t)v.je * v.ke, (size_t)v.ie * v.ke, (size_t)v.ie * v.je};
        for (int f6 = 0; f6 < 6; ++f6)
            for (int which = 0; which < 2; ++which) {
                void* raw = nullptr;
                HIPCHK(hipMalloc(&raw, sizeof(double) * nf[f6 / 2]));
                HIPCHK(hipMemsetAsync(raw, 0, sizeof(double) * nf[f6 / 2], g_stream));
                b->allocs.push_back(raw);
                (which ? b->v.bvt : b->v.bmt)[f6] = (double*)raw;
            }
        invalidate_comm_level(level);   // the device block table holds copies of BlkView
    } else if (v.nw > 5) {
        // a new registration: the faces that carry no subface any more must read bmt = bvt = 0 (the merged turbulence treatment
        // writes the cells of the subfaces only)
        const size_t nf[3] = {(size_t)v.je * v.ke, (size_t)v.ie * v.ke, (size_t)v.ie * v.je};
        for (int f6 = 0; f6 < 6; ++f6) {
            HIPCHK(hipMemsetAsync(b->v.bmt[f6], 0, sizeof(double) * nf[f6 / 2], g_stream));
            HIPCHK(hipMemsetAsync(b->v.bvt[f6], 0, sizeof(double) * nf[f6 / 2], g_stream));
        }
    }
    return 0;
}

// ---- launch plan of the boundary subfaces of a level (kernels_bc.hip): every subface of every block in one device
// array; `flow` lists the launches of applyAllBC in the reference's order, `ordinal` the r-th subfaces of all blocks
struct BcPlan {
    BcEntry* d_ent = nullptr;
    int* d_order = nullptr;
    std::vector<BcPhase> flow, ordinal;
    BcPhase wall = {0, 0, 0, 0};   // the viscous subfaces (wall stress storage), any order
    long maxFace = 0;
    bool anyEulerWall = false;
    int nent = 0;
};
static std::map<int, BcPlan> g_bcplan;

static void bc_plan_drop(int level)
{
    auto it = g_bcplan.find(level);
    if (it == g_bcplan.end()) return;
    if (it->second.d_ent) (void)hipFree(it->second.d_ent);
    if (it->second.d_order) (void)hipFree(it->second.d_order);
    g_bcplan.erase(it);
}

static void bc_plan_drop_all()
{
    std::vector<int> levels;
    for (auto& kv : g_bcplan) levels.push_back(kv.first);
    for (int l : levels) bc_plan_drop(l);
}

static int bc_plan(int level, BcPlan** out)
{
    auto it = g_bcplan.find(level);
    if (it != g_bcplan.end()) { *out = &it->second; return 0; }
    BcPlan pl;
    std::vector<BcEntry> ent;
    struct Ref { int first, n, nVisc; };      // entries of one block: ent[first .. first+n)
    std::vector<Ref> blocks;
    for (auto& kv : g_blocks) {
        if (std::get<0>(kv.first) != level || std::get<1>(kv.first) != 1) continue;
        Block* b = kv.second;
        if (b->bc.empty()) continue;
        Ref r = {(int)ent.size(), (int)b->bc.size(), b->nViscBocos};
        for (auto& f : b->bc) {
            BcEntry e;
            e.slot = std::get<2>(kv.first); e.pad = 0; e.f = f;
            ent.push_back(e);
            pl.anyEulerWall = pl.anyEulerWall || f.type == ADFLOW_BC_EULERWALL;
        }
        blocks.push_back(r);
        const BlkView& v = b->v;
        pl.maxFace = std::max(pl.maxFace, std::max(std::max((long)v.je * v.ke, (long)v.ie * v.ke), (long)v.ie * v.je));
    }
    pl.nent = (int)ent.size();
    std::vector<int> order;
    auto cells = [&](int e) { const BcFaceDev& f = ent[e].f; return (long)(f.icEnd - f.icBeg + 1) * (f.jcEnd - f.jcBeg + 1); };
    // kinds in the order of applyAllBC_block (BCRoutines.F90:75-216); walls only among the first nViscBocos subfaces
    auto match = [&](int kind, const BcFaceDev& f, bool inVisc) {
        switch (kind) {
        case BCP_SYMM1: case BCP_SYMM2: return f.type == ADFLOW_BC_SYMM;
        case BCP_WALL_ADIABATIC: return inVisc && f.type == ADFLOW_BC_NSWALL_ADIABATIC;
        case BCP_WALL_ISOTHERMAL: return inVisc && f.type == ADFLOW_BC_NSWALL_ISOTHERMAL;
        case BCP_FARFIELD: return f.type == ADFLOW_BC_FARFIELD;
        case BCP_EXTRAP: return f.type == ADFLOW_BC_EXTRAP || f.type == ADFLOW_BC_SUPERSONIC_OUTFLOW;
        case BCP_EULERWALL: return f.type == ADFLOW_BC_EULERWALL;
        case BCP_SUPERSONIC_INFLOW: return f.type == ADFLOW_BC_SUPERSONIC_INFLOW;
        case BCP_SYMMPOLAR1: case BCP_SYMMPOLAR2: return f.type == ADFLOW_BC_SYMM_POLAR;
        case BCP_SUBSONIC_OUTFLOW: return f.type == ADFLOW_BC_SUBSONIC_OUTFLOW || f.type == ADFLOW_BC_MASSBLEED_OUTFLOW;
        case BCP_SUBSONIC_INFLOW: return f.type == ADFLOW_BC_SUBSONIC_INFLOW;
        default: return true;
        }
    };
    auto add_kind = [&](int kind, std::vector<BcPhase>& dst) {
        for (int r = 0;; ++r) {       // r-th matching subface of every block: never two subfaces of one block together
            BcPhase ph = {kind, (int)order.size(), 0, 0};
            for (auto& rb : blocks) {
                int seen = 0;
                for (int m = 0; m < rb.n; ++m)
                    if (match(kind, ent[rb.first + m].f, m < rb.nVisc)) {
                        if (seen == r) { order.push_back(rb.first + m); ph.count++; ph.maxCells = std::max(ph.maxCells, cells(rb.first + m)); break; }
                        ++seen;
                    }
            }
            if (ph.count == 0) break;
            dst.push_back(ph);
        }
    };
    for (int kind : {BCP_SYMM1, BCP_SYMM2, BCP_SYMMPOLAR1, BCP_SYMMPOLAR2, BCP_WALL_ADIABATIC, BCP_WALL_ISOTHERMAL, BCP_FARFIELD,
                     BCP_SUBSONIC_OUTFLOW, BCP_SUBSONIC_INFLOW, BCP_EXTRAP, BCP_EULERWALL, BCP_SUPERSONIC_INFLOW})
        add_kind(kind, pl.flow);
    add_kind(BCP_ORDINAL, pl.ordinal);
    pl.wall.first = (int)order.size();
    for (int e = 0; e < (int)ent.size(); ++e)
        if (ent[e].f.tauq) {
            order.push_back(e);
            pl.wall.count++;
            // (the launch over the wall faces also covers the (n1 + 1) (n2 + 1) nodes of their plane: k_wall_node_grad)
            const BcFaceDev& f = ent[e].f;
            pl.wall.maxCells = std::max(pl.wall.maxCells, (long)(f.icEnd - f.icBeg + 2) * (f.jcEnd - f.jcBeg + 2));
        }
    if (pl.nent > 0) {
        HIPCHK(hipMalloc((void**)&pl.d_ent, sizeof(BcEntry) * ent.size()));
        HIPCHK(hipMemcpy(pl.d_ent, ent.data(), sizeof(BcEntry) * ent.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMalloc((void**)&pl.d_order, sizeof(int) * order.size()));
        HIPCHK(hipMemcpy(pl.d_order, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice));
    }
    g_bcplan[level] = pl;
    *out = &g_bcplan[level];
    return 0;
}

// bcTurbTreatment of every block of the level with registered subfaces
static int turb_bc_treatment_enqueue(int level, const KParams& kp)
{
    BcPlan* pl;
    if (bc_plan(level, &pl)) return 1;
    if (pl->nent == 0) return 0;
    LevelTab t;
    if (level_tab(level, &t)) return 1;
    launch_turb_bc_treatment(t.tab, t.n, pl->maxFace, pl->d_ent, pl->d_order, pl->ordinal, kp, g_stream);
    return 0;
}

// applyAllTurbBCThisBlock(secondHalo) of every block of the level
static int turb_bc_apply_enqueue(int level, const KParams& kp, int secondHalo)
{
    BcPlan* pl;
    if (bc_plan(level, &pl)) return 1;
    if (pl->nent == 0) return 0;
    LevelTab t;
    if (level_tab(level, &t)) return 1;
    launch_apply_turb_bc(t.tab, pl->d_ent, pl->d_order, pl->ordinal, kp, secondHalo, g_stream);
    return 0;
}

// bcTurbTreatment + applyAllTurbBCThisBlock(secondHalo) for the blocks of `level` with registered subfaces
static int apply_turb_bc_enqueue(int level, int secondHalo)
{
    if (g_opts.equations != ADFLOW_RANS) return 0;
    KParams kp = make_kparams(level, 1.0, 0);
    if (turb_bc_treatment_enqueue(level, kp)) return 1;
    return turb_bc_apply_enqueue(level, kp, secondHalo);
}

// both, in the order of blocketteRes (blockette.F90:228-244): turbulence first
static int apply_turb_and_flow_bc_enqueue(int level, int secondHalo, bool turbBC)
{
    if (turbBC && apply_turb_bc_enqueue(level, secondHalo)) return 1;
    return apply_bc_enqueue(level, secondHalo);
}

// applyAllBC (BCRoutines.F90:15-54) for the blocks of `level` that registered subfaces
static int apply_bc_enqueue(int level, int secondHalo)
{
    BcPlan* pl;
    if (bc_plan(level, &pl)) return 1;
    if (pl->nent == 0) return 0;
    KParams kp = make_kparams(level, 1.0, 0);
    if (pl->anyEulerWall && g_opts.eulerWallBCTreatment == ADFLOW_WALLBC_QUADRATIC)
        return fail("eulerWallBCTreatment=%d: bcEulerWall has no quadratic extrapolation (1 constant, 2 linear, 4 normal momentum)",
                    g_opts.eulerWallBCTreatment);
    if (pl->anyEulerWall && g_opts.eulerWallBCTreatment == ADFLOW_WALLBC_NORMAL_MOMENTUM && kp.fineGrid) {
        bool moving = false;
        for_level(level, [&](Block* b) { moving = moving || b->v.sFace || b->v.moving; return 0; });
        if (moving) return fail("eulerWallBCTreatment = normal momentum on moving blocks needs the cell-centre grid velocity (not mirrored)");
    }
    LevelTab t;
    if (level_tab(level, &t)) return 1;
    launch_apply_all_bc(t.tab, pl->d_ent, pl->d_order, pl->flow, kp, secondHalo, g_opts.eulerWallBCTreatment,
                        g_opts.viscWallBCTreatment, g_opts.outflowTreatment, g_opts.hScalingInlet, g_stream);
    return for_level(level, [&](Block* b) {
        if (!b->bc.empty()) b->ss_valid = false;
        return 0;
    });
}

// bcTurbTreatment_d + applyAllTurbBCThisBlock_d + applyAllBC_block_d on the dual arrays of the level (kernels_ad.hip)
static int ad_apply_bc_enqueue(int level, const KParams& kp, bool turbBC)
{
    BcPlan* pl;
    if (bc_plan(level, &pl)) return 1;
    if (pl->nent == 0) return 0;
    if (pl->anyEulerWall && g_opts.eulerWallBCTreatment == ADFLOW_WALLBC_QUADRATIC)
        return fail("eulerWallBCTreatment=%d: bcEulerWall has no quadratic extrapolation", g_opts.eulerWallBCTreatment);
    LevelTab t;
    if (level_tab(level, &t)) return 1;
    if (turbBC) ad_launch_turb_bc(g_ad_tab, t.n, pl->maxFace, pl->d_ent, pl->d_order, pl->ordinal, kp, 1, g_stream);
    // applyAllBC_block_d (src/adjoint/outputForward/BCExtra_d.F90:10-139) applies the kinds that were differentiated: symmetry,
    // polar symmetry, viscous walls, farfield, subsonic out- / inflow, Euler wall.  Extrapolation and the supersonic kinds are NOT
    // in it: their halos keep the values of the last primal boundary-condition pass with a zero derivative.  Reproduced.
    std::vector<BcPhase> flow;
    for (const BcPhase& ph : pl->flow)
        if (ph.kind != BCP_EXTRAP && ph.kind != BCP_SUPERSONIC_INFLOW) flow.push_back(ph);
    ad_launch_apply_all_bc(g_ad_tab, pl->d_ent, pl->d_order, flow, kp, 1, g_opts.eulerWallBCTreatment, g_opts.viscWallBCTreatment,
                           g_opts.outflowTreatment, g_opts.hScalingInlet, g_stream);
    return 0;
}

static bool has_wall_subfaces(int level)
{
    BcPlan* pl;
    if (bc_plan(level, &pl)) return false;
    return pl->wall.count > 0;
}

// viscSubface(:)%tau / %q of every viscous subface of the level, from the nodal gradients the viscous kernels just used
static int wall_stress_enqueue(int level, const KParams& kp, bool formGrad)
{
    BcPlan* pl;
    if (bc_plan(level, &pl)) return 1;
    if (pl->wall.count == 0) return 0;
    LevelTab t;
    if (level_tab(level, &t)) return 1;
    launch_wall_stress(t.tab, pl->d_ent, pl->d_order, pl->wall, kp, formGrad, g_stream);
    return 0;
}

int adflow_gpu_download_wall_stress(int nn, int level, int sps, int mm, double* tau, double* q)
{
    Block* b = find_block(nn, level, sps);
    if (!b) return fail("block (%d,%d,%d) not registered", nn, level, sps);
    if (mm < 1 || mm > b->nViscBocos || mm > (int)b->bc.size()) return fail("download_wall_stress: subface %d is not a viscous subface of block %d", mm, nn);
    const BcFaceDev& f = b->bc[mm - 1];
    int r[4];
    bc_owned_range(f.faceID, f.icBeg, f.icEnd, f.jcBeg, f.jcEnd, b->v.il, b->v.jl, b->v.kl, r);
    const long no = (long)std::max(0, r[1] - r[0] + 1) * std::max(0, r[3] - r[2] + 1);
    if (no == 0 || !f.tauq) return 0;
    HIPCHK(hipStreamSynchronize(g_stream));
    if (tau) HIPCHK(hipMemcpy(tau, f.tauq, sizeof(double) * 6 * no, hipMemcpyDeviceToHost));
    if (q) HIPCHK(hipMemcpy(q, f.tauq + 6 * no, sizeof(double) * 3 * no, hipMemcpyDeviceToHost));
    return 0;
}

// setCorrectionsCoarseHalos of every coarse block with subfaces (multiGrid.F90:472)
static int bc_coarse_corrections_enqueue(int coarseLevel, double fact)
{
    BcPlan* pl;
    if (bc_plan(coarseLevel, &pl)) return 1;
    if (pl->nent == 0) return 0;
    LevelTab t;
    if (level_tab(coarseLevel, &t)) return 1;
    launch_bc_coarse_corrections(t.tab, pl->d_ent, pl->d_order, pl->ordinal, fact, g_stream);
    return 0;
}

int adflow_gpu_apply_all_bc(int level, int secondHalo)
{
    if (need_ready()) return 1;
    if (apply_bc_enqueue(level, secondHalo)) return 1;
    return sync_and_check();
}

int adflow_gpu_comm_unique_id(void* id128)
{
#ifndef ADFLOW_NO_RCCL
    if (!id128) return fail("null id buffer");
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return 0;
#else
    (void)id128;
    return fail("built without RCCL");
#endif
}

int adflow_gpu_comm_init(int rank, int nranks, const void* id128)
{
#ifndef ADFLOW_NO_RCCL
    if (g_device < 0) return fail("adflow_gpu_init has not been called");
    if (!id128) return fail("null id buffer");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    NCCLCHK(ncclCommInitRank(&g_nccl, nranks, id, rank));
    g_rank = rank;
    g_nranks = nranks;
    g_self_rank = rank;
    return 0;
#else
    (void)rank; (void)nranks; (void)id128;
    return fail("built without RCCL");
#endif
}

int adflow_gpu_comm_info(int* rank, int* nranks, int* commCount, int* commUserRank)
{
    int cnt = -1, ur = -1;
#ifndef ADFLOW_NO_RCCL
    if (rank) *rank = g_rank;
    if (nranks) *nranks = g_nranks;
    if (g_nccl) {
        NCCLCHK(ncclCommCount(g_nccl, &cnt));
        NCCLCHK(ncclCommUserRank(g_nccl, &ur));
    }
#else
    if (rank) *rank = 0;
    if (nranks) *nranks = 1;
#endif
    if (commCount) *commCount = cnt;
    if (commUserRank) *commUserRank = ur;
    return 0;
}

int adflow_gpu_comm_register(int level, int nLayers, const adflow_comm_pattern* p)
{
    ++g_state_gen;
    if (g_device < 0) return fail("adflow_gpu_init has not been called");
    if (!p) return fail("null comm pattern");
    if (nLayers < 0 || nLayers > 2) return fail("nLayers must be 1 or 2 (cell halos) or 0 (the node pattern of exchangeCoor)");
    auto key = std::make_pair(level, nLayers);
    if (g_comm.count(key)) {
        drop_comm_lists(g_comm[key]);     // a re-registration also drops the periodic transformations
    }
    CommPattern cp;
    cp.present = true;
    const int nc = p->ncopy;
    if (nc < 0 || p->nProcSend < 0 || p->nProcRecv < 0) return fail("negative count in comm pattern");
    if (nc > 0) {
        cp.h_donorBlock.assign(p->donorBlock, p->donorBlock + nc);
        cp.h_haloBlock.assign(p->haloBlock, p->haloBlock + nc);
        cp.h_donorIdx.assign(p->donorIndices, p->donorIndices + 3 * (size_t)nc);
        cp.h_haloIdx.assign(p->haloIndices, p->haloIndices + 3 * (size_t)nc);
    }
    if (p->nProcSend > 0) {
        cp.h_sendProc.assign(p->sendProc, p->sendProc + p->nProcSend);
        cp.h_nsendCum.assign(p->nsendCum, p->nsendCum + p->nProcSend + 1);
        const int nt = cp.h_nsendCum[p->nProcSend];
        cp.h_sendBlock.assign(p->sendBlock, p->sendBlock + nt);
        cp.h_sendIdx.assign(p->sendIndices, p->sendIndices + 3 * (size_t)nt);
    }
    if (p->nProcRecv > 0) {
        cp.h_recvProc.assign(p->recvProc, p->recvProc + p->nProcRecv);
        cp.h_nrecvCum.assign(p->nrecvCum, p->nrecvCum + p->nProcRecv + 1);
        const int nt = cp.h_nrecvCum[p->nProcRecv];
        cp.h_recvBlock.assign(p->recvBlock, p->recvBlock + nt);
        cp.h_recvIdx.assign(p->recvIndices, p->recvIndices + 3 * (size_t)nt);
    }
    g_comm[key] = cp;
    return 0;
}

int adflow_gpu_comm_register_periodic(int level, int nLayers, int nPeriodic, const adflow_periodic_data* pd)
{
    ++g_state_gen;
    auto it = g_comm.find(std::make_pair(level, nLayers));
    if (it == g_comm.end()) return fail("comm_register_periodic: no pattern registered for level %d, nLayers %d", level, nLayers);
    if (nPeriodic < 0 || (nPeriodic > 0 && !pd)) return fail("comm_register_periodic: nPeriodic=%d", nPeriodic);
    CommPattern& cp = it->second;
    if (g_stream) (void)hipStreamSynchronize(g_stream);
    drop_comm_lists(cp);
    cp.periodic.clear();
    for (int m = 0; m < nPeriodic; ++m) {
        CommPattern::Periodic q;
        for (int a = 0; a < 9; ++a) q.rotMatrix[a] = pd[m].rotMatrix[a];
        for (int a = 0; a < 3; ++a) { q.rotCenter[a] = pd[m].rotCenter[a]; q.translation[a] = pd[m].translation[a]; }
        const int n = pd[m].nHalos;
        if (n < 0 || (n > 0 && (!pd[m].block || !pd[m].indices))) return fail("comm_register_periodic: entry %d: nHalos=%d", m + 1, n);
        q.h_block.assign(pd[m].block, pd[m].block + n);
        q.h_idx.assign(pd[m].indices, pd[m].indices + (size_t)3 * n);
        cp.periodic.push_back(q);
    }
    return 0;
}

int adflow_gpu_halo_local_copy(int level, int nLayers, int varStart, int varEnd, int commPressure, int commVisc)
{
    if (need_ready()) return 1;
    CommPattern* cp;
    if (build_comm(level, nLayers, &cp)) return 1;
    unsigned mask; int nvar;
    if (halo_mask(varStart, varEnd, commPressure, commVisc, &mask, &nvar)) return 1;
    launch_halo_copy(g_tab[level], cp->local.blkA, cp->local.offA, cp->local.blkB, cp->local.offB, cp->local.n, mask, g_stream);
    for_level(level, [&](Block* b) { b->ss_valid = false; return 0; });
    return sync_and_check();
}

int adflow_gpu_halo_slot_info(int level, int nLayers, int isSend, int islot, int* peer, int* count)
{
    CommPattern* cp;
    if (build_comm(level, nLayers, &cp)) return 1;
    std::vector<CommList>& v = isSend ? cp->sends : cp->recvs;
    if (islot < 0 || islot >= (int)v.size()) {
        if (peer) *peer = -1;
        if (count) *count = 0;
        return 0;   // past the last slot: count 0
    }
    if (peer) *peer = v[islot].peer;
    if (count) *count = v[islot].n;
    return 0;
}

int adflow_gpu_halo_pack(int level, int nLayers, int islot, int varStart, int varEnd, int commPressure, int commVisc, double* buf)
{
    if (need_ready()) return 1;
    CommPattern* cp;
    if (build_comm(level, nLayers, &cp)) return 1;
    if (islot < 0 || islot >= (int)cp->sends.size()) return fail("send slot %d out of range", islot);
    unsigned mask; int nvar;
    if (halo_mask(varStart, varEnd, commPressure, commVisc, &mask, &nvar)) return 1;
    CommList& l = cp->sends[islot];
    launch_halo_pack(g_tab[level], l.blkA, l.offA, l.n, mask, l.buf, g_stream);
    HIPCHK(hipMemcpyAsync(buf, l.buf, sizeof(double) * (size_t)nvar * l.n, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    return 0;
}

int adflow_gpu_halo_unpack(int level, int nLayers, int islot, int varStart, int varEnd, int commPressure, int commVisc,
                           const double* buf)
{
    if (need_ready()) return 1;
    CommPattern* cp;
    if (build_comm(level, nLayers, &cp)) return 1;
    if (islot < 0 || islot >= (int)cp->recvs.size()) return fail("recv slot %d out of range", islot);
    unsigned mask; int nvar;
    if (halo_mask(varStart, varEnd, commPressure, commVisc, &mask, &nvar)) return 1;
    CommList& l = cp->recvs[islot];
    HIPCHK(hipMemcpyAsync(l.buf, buf, sizeof(double) * (size_t)nvar * l.n, hipMemcpyHostToDevice, g_stream));
    launch_halo_unpack(g_tab[level], l.blkA, l.offA, l.n, mask, l.buf, g_stream);
    HIPCHK(hipStreamSynchronize(g_stream));
    for_level(level, [&](Block* b) { b->ss_valid = false; return 0; });
    return 0;
}

static int comm_exchange_enqueue(CommPattern* cp, BlkView* tab, unsigned mask, int nvar);

// exchangePressureEarly (smoothers.F90:363-365, 674-676, multiGrid.F90:602-604): with the normal-momentum treatment of inviscid
// walls bcEulerWall reads the pressure of the first halo layer of the neighbouring blocks, so the reference exchanges the pressure
// alone -- whalo1(currentLevel, 1, 0, .true., .false., .false.) -- BEFORE applyAllBC, on the fine grid only
static int early_pressure_exchange_enqueue(int level)
{
    if (!g_opts.exchangePressureEarly || level > g_opts.groundLevel) return 0;
    if (!g_comm.count(std::make_pair(level, 1)))
        return fail("exchangePressureEarly: the 1-layer cell pattern of level %d (commPatternCell_1st / internalCell_1st) is not registered",
                    level);
    return halo_exchange_enqueue(level, 1, 0, 1, 0, 1);
}

static int halo_exchange_enqueue(int level, int varStart, int varEnd, int commPressure, int commVisc, int nLayers)
{
    CommPattern* cp;
    if (build_comm(level, nLayers, &cp)) return 1;
    unsigned mask; int nvar;
    if (halo_mask(varStart, varEnd, commPressure, commVisc, &mask, &nvar)) return 1;
    if (nvar == 0) return 0;
    if (comm_exchange_enqueue(cp, g_tab[level], mask, nvar)) return 1;
    return halo_exchange_close(level, varStart, varEnd, commPressure, nLayers);
}

// whalo2 closes by recomputing the total energy of the owned cells from p when both travelled (haloExchange.F90:178-196)
static int halo_exchange_close(int level, int varStart, int varEnd, int commPressure, int nLayers)
{
    const bool bothPAndE = commPressure && varStart <= 5 && varEnd >= 5;
    int nTodo = 0, nBlk = 0;
    for_level(level, [&](Block* b) {
        ++nBlk;
        if (nLayers == 2 && bothPAndE && !b->etot_consistent) ++nTodo;
        return 0;
    });
    if (nTodo > 0 && nTodo == nBlk) {
        LevelTab t;
        if (level_tab(level, &t)) return 1;
        launch_etot_owned_level(t.tab, t.n, t.nx, t.ny, t.nz, g_opts.gammaConstant, g_stream,
                                (g_etot_flag_level == level) ? g_floor_flag_dev : nullptr);
    }
    for_level(level, [&](Block* b) {
        // the exchange never touches owned cells: when their rhoE was produced by
        // computeEtotBlock already (stage update, or a previous whalo2) the pass is an identity
        if (nLayers == 2 && bothPAndE && !b->etot_consistent) {
            if (nTodo != nBlk) launch_etot_owned(b->v, g_opts.gammaConstant, g_stream);
            b->etot_consistent = true;
        }
        b->ss_valid = false;
        return 0;
    });
    return 0;
}

// pack -> grouped RCCL send/recv (own queue) || same-GPU copies -> unpack of the variables in `mask` over one pattern, in two halves:
// begin = everything up to the messages in flight and the same-GPU copies, end = the wait for the group, the unpacks and the periodic
// transformations.  The evaluation split around the exchange (block_res_split_enqueue) puts the halo-free tiles between the two.
static int comm_exchange_begin(CommPattern* cp, BlkView* tab, unsigned mask, int nvar, bool* remoteOut)
{
    // pack every outgoing message on the compute queue, then ONE grouped RCCL send/recv over xGMI on the communication queue;
    // the same-GPU copies (they read owned cells and write halos no message touches) run on the compute queue WHILE the
    // messages are in flight, the unpacks wait for the group (haloExchange.F90:553-719 does the same with isend / irecv /
    // local copy / waitany)
    for (auto& l : cp->sends) launch_halo_pack(tab, l.blkA, l.offA, l.n, mask, l.buf, g_stream);
    const bool remote = !cp->sends.empty() || !cp->recvs.empty();
    *remoteOut = remote;
    if (remote) {
#ifndef ADFLOW_NO_RCCL
        if (!g_nccl) return fail("halo exchange needs other ranks but adflow_gpu_comm_init was not called");
        hipStream_t sx = g_overlap ? g_streamX : g_stream;
        if (g_overlap) {
            HIPCHK(hipEventRecord(g_evPack, g_stream));
            HIPCHK(hipStreamWaitEvent(g_streamX, g_evPack, 0));
        }
        NCCLCHK(ncclGroupStart());
        for (auto& l : cp->sends)
            if (l.n > 0) NCCLCHK(ncclSend(l.buf, (size_t)nvar * l.n, ncclDouble, l.peer, g_nccl, sx));
        for (auto& l : cp->recvs)
            if (l.n > 0) NCCLCHK(ncclRecv(l.buf, (size_t)nvar * l.n, ncclDouble, l.peer, g_nccl, sx));
        NCCLCHK(ncclGroupEnd());
        if (g_overlap) HIPCHK(hipEventRecord(g_evComm, g_streamX));
#else
        return fail("built without RCCL: use adflow_gpu_halo_pack/unpack with an external transport");
#endif
    }
    launch_halo_copy(tab, cp->local.blkA, cp->local.offA, cp->local.blkB, cp->local.offB, cp->local.n, mask, g_stream);
    return 0;
}

static int comm_exchange_end(CommPattern* cp, BlkView* tab, unsigned mask, bool remote)
{
#ifndef ADFLOW_NO_RCCL
    if (remote && g_overlap) HIPCHK(hipStreamWaitEvent(g_stream, g_evComm, 0));
#endif
    (void)remote;
    for (auto& l : cp->recvs) launch_halo_unpack(tab, l.blkA, l.offA, l.n, mask, l.buf, g_stream);
    // periodic transformations of the halos that crossed a periodic interface: coordinates for the node pattern,
    // velocities when all three travelled (haloExchange.F90:456-457)
    const bool coor = (mask & (7u << 11)) != 0;
    const bool vel = (mask & 0xEu) == 0xEu;
    if (coor || vel)
        for (auto& pd : cp->periodic)
            launch_periodic(tab, pd.list.blkA, pd.list.offA, pd.list.n, pd.rotMatrix, pd.rotCenter, pd.translation, coor ? 1 : 0, g_stream);
    return 0;
}

static int comm_exchange_enqueue(CommPattern* cp, BlkView* tab, unsigned mask, int nvar)
{
    bool remote = false;
    if (comm_exchange_begin(cp, tab, mask, nvar, &remote)) return 1;
    return comm_exchange_end(cp, tab, mask, remote);
}

// exchangeCoor (haloExchange.F90:2456-2640): the three coordinates over the node pattern (nLayers key 0)
int adflow_gpu_exchange_coor(int level)
{
    if (need_ready()) return 1;
    CommPattern* cp;
    if (build_comm(level, 0, &cp)) return 1;
    if (comm_exchange_enqueue(cp, g_tab[level], 7u << 11, 3)) return 1;
    for_level(level, [&](Block* b) { b->face_vectors_valid = false; return 0; });
    return sync_and_check();
}

// xhalo_block (adjointExtra.F90:365-599) of every block of the level
int adflow_gpu_xhalo(int level)
{
    if (need_ready()) return 1;
    LevelTab t;
    if (level_tab(level, &t)) return 1;
    launch_xhalo_level(t.tab, t.n, t.nx, t.ny, t.nz, g_stream);
    BcPlan* pl;
    if (bc_plan(level, &pl)) return 1;
    for (const BcPhase& ph : pl->ordinal) launch_xhalo_symm(t.tab, pl->d_ent, pl->d_order, ph, g_stream);
    for_level(level, [&](Block* b) { b->face_vectors_valid = false; return 0; });
    return sync_and_check();
}

int adflow_gpu_halo_exchange(int level, int varStart, int varEnd, int commPressure, int commVisc, int nLayers)
{
    if (need_ready()) return 1;
    if (halo_exchange_enqueue(level, varStart, varEnd, commPressure, commVisc, nLayers)) return 1;
    return sync_and_check();
}

// ----------------------------------------------------------------- smoothers
// one stage of either smoother after dw holds the scaled update: state update,
// boundary-condition hook, halo exchange (smoothers.F90:292-380, 600-691)
// updated: the state update ran inside the kernel that completed the increment (D-ADI k sweep)
static int finish_stage(int level, const KParams& kp, double scale, int fromWn, bool updated = false)
{
    LevelTab t;
    if (level_tab(level, &t)) return 1;
    if (!updated) launch_stage_update_level(t.tab, t.n, t.nx, t.ny, t.nz, kp, scale, fromWn, g_stream);
    int rc = for_level(level, [&](Block* b) {
        b->ss_valid = false;
        b->etot_consistent = true;
        return 0;
    });
    if (rc) return rc;
    const int secondHalo = (level <= g_opts.groundLevel);
    if (early_pressure_exchange_enqueue(level)) return 1;
    if (apply_bc_enqueue(level, secondHalo)) return 1;
    if (g_bc_callback) {
        HIPCHK(hipStreamSynchronize(g_stream));
        g_bc_callback(level, secondHalo);
    }
    const int nLayers = secondHalo ? 2 : 1;
    if (g_comm.count(std::make_pair(level, nLayers)))
        if (halo_exchange_enqueue(level, 1, 5, 1, 1, nLayers)) return 1;
    return 0;
}

static bool smooth_residual(int rkStage)
{
    if (g_opts.resAveraging == ADFLOW_RESAVG_NEVER) return false;
    if (g_opts.resAveraging == ADFLOW_RESAVG_ALWAYS) return true;
    return (rkStage % 2) == 1;
}

int adflow_gpu_rk_smooth(int level)
{
    if (need_ready()) return 1;
    if (g_opts.smoother != ADFLOW_RUNGE_KUTTA) return fail("adflow_gpu_rk_smooth called with smoother=%d", g_opts.smoother);
    LevelTab t;
    if (level_tab(level, &t)) return 1;
    launch_rk_save_level(t.tab, t.n, t.nx, t.ny, t.nz, g_stream);
    const int nst = g_opts.nRKStages;
    for (int stage = 1; stage <= nst; ++stage) {
        KParams kp = make_kparams(level, 1.0, 1);
        const double scale = (g_opts.lowSpeedPreconditioner ? 0.8 : 1.0) * kp.cfl * g_opts.etaRK[stage - 1];   // smoothers.F90:202
        if (level_tab(level, &t)) return 1;      // a BC callback of the previous stage may have rebuilt the device table
        if (smooth_residual(stage)) {
            if (res_averaging_level(level, kp, scale)) return 1;
            if (finish_stage(level, kp, 0.0, 1)) return 1;
        } else {
            if (finish_stage(level, kp, scale, 1)) return 1;
        }
        if (stage < nst) {
            // residual of the next stage: rkStage = stage -> rFil = cdisRK(stage+1)
            KParams kr = make_kparams(level, g_opts.cdisRK[stage], 1);
            if (enqueue_flow_residual(level, kr, false, true, false)) return 1;
        }
    }
    return sync_and_check();
}

int adflow_gpu_dadi_smooth(int level)
{
    if (need_ready()) return 1;
    if (g_opts.smoother != ADFLOW_DADI) return fail("adflow_gpu_dadi_smooth called with smoother=%d", g_opts.smoother);
    const int nsub = (g_opts.groundLevel == 1) ? std::max(1, (int)g_opts.nSubiterations) : 1;
    for (int it = 1; it <= nsub; ++it) {
        KParams kp = make_kparams(level, 1.0, 0);
        LevelTab t;
        if (level_tab(level, &t)) return 1;
        const bool fuseUpd = !smooth_residual(0) && g_dadi_upd;               // (tuning "dadi_upd")
        launch_dadi_level(t.tab, t.n, t.nx, t.ny, t.nz, kp, g_stream, fuseUpd);
        if (smooth_residual(0) && res_averaging_level(level, kp)) return 1;   // rkStage stays 0 under DADI
        if (finish_stage(level, kp, 0.0, 0, fuseUpd)) return 1;
        if (it < nsub) {
            if (enqueue_flow_residual(level, kp)) return 1;
        }
    }
    return sync_and_check();
}

// ---------------------------------------------------------------- multigrid
static int for_level_pairs(int fineLevel, const std::function<int(Block*, Block*)>& fn)
{
    bool any = false;
    for (auto& kv : g_blocks) {
        if (std::get<0>(kv.first) != fineLevel) continue;
        Block* c = find_block(std::get<2>(kv.first), fineLevel + 1, std::get<1>(kv.first));
        if (!c) return fail("block %d has no level-%d counterpart", std::get<2>(kv.first), fineLevel + 1);
        any = true;
        int rc = fn(kv.second, c);
        if (rc) return rc;
    }
    if (!any) return fail("no block registered on level %d", fineLevel);
    return 0;
}

static int exchange_if_registered(int level, int nLayers)
{
    if (g_comm.count(std::make_pair(level, nLayers))) return halo_exchange_enqueue(level, 1, 5, 1, 1, nLayers);
    return 0;
}

static double rfil_stage0(int* fwMode)
{
    *fwMode = (g_opts.smoother == ADFLOW_RUNGE_KUTTA) ? 1 : 0;
    return (g_opts.smoother == ADFLOW_RUNGE_KUTTA) ? g_opts.cdisRK[0] : 1.0;
}

static int transfer_to_coarse_enqueue(int level)
{
    const int cl = level + 1;
    int fwMode;
    const double rFil = rfil_stage0(&fwMode);
    // residual of the fine level with rkStage = 0 (multiGrid.F90:62-70)
    KParams kf = make_kparams(level, rFil, fwMode);
    kf.onlyRadii = 1;
    int rc = time_step_level(level, kf);
    if (rc) return rc;
    if (enqueue_flow_residual(level, kf)) return 1;
    // restriction + closures on the coarse level
    KParams kc = make_kparams(cl, rFil, fwMode);
    rc = for_level_pairs(level, [&](Block* f, Block* c) {
        if (!c->v.mgIFine || !c->v.mgIWeight) return fail("coarse block has no mgIFine/mgIWeight maps");
        if (!c->geom_uploaded) return fail("geometry of a level-%d block has not been uploaded", cl);
        c->ss_valid = false;
        c->etot_consistent = true;
        return 0;
    });
    if (rc) return rc;
    LevelTab tf, tc;
    if (level_tab(level, &tf) || level_tab(cl, &tc)) return 1;
    launch_restrict_level(tc.tab, tf.tab, tc.n, tc.nx, tc.ny, tc.nz, kc, g_stream);
    launch_corner_row_halos_level(tc.tab, tc.n, kc, g_stream);      // setCornerRowHalos(nwf) (multiGrid.F90:229)
    // applyAllBC(.false.) ; whalo1 (multiGrid.F90:236-241)
    if (apply_bc_enqueue(cl, 0)) return 1;
    if (g_bc_callback) {
        HIPCHK(hipStreamSynchronize(g_stream));
        g_bc_callback(cl, 0);
    }
    if (exchange_if_registered(cl, 1)) return 1;
    // time step, entry state, coarse residual from zero, forcing term (multiGrid.F90:246-320)
    kc.onlyRadii = 0;
    rc = time_step_level(cl, kc);
    if (rc) return rc;
    if (level_tab(cl, &tc)) return 1;       // BC registration may have rebuilt the table
    launch_store_entry_state_level(tc.tab, tc.n, tc.nx, tc.ny, tc.nz, g_stream);
    KParams kz = kc;
    kz.coarseInit = 0;
    if (enqueue_flow_residual(cl, kz)) return 1;
    launch_forcing_level(tc.tab, tc.n, tc.nx, tc.ny, tc.nz, g_opts.fcoll, g_stream);
    return 0;
}

static int transfer_to_fine_enqueue(int level)
{
    KParams kf = make_kparams(level, 1.0, 0);
    LevelTab tf, tc;
    if (level_tab(level, &tf) || level_tab(level + 1, &tc)) return 1;
    launch_corrections_level(tc.tab, tc.n, tc.nx, tc.ny, tc.nz, g_stream);
    int rc = for_level_pairs(level, [&](Block* f, Block* c) {
        if (!f->v.mgICoarse) return fail("fine block has no mgICoarse map");
        f->ss_valid = false;
        f->etot_consistent = true;
        return 0;
    });
    if (rc) return rc;
    // setCorrectionsCoarseHalos (multiGrid.F90:472): fact = 0, mgBoundCorr = bcDirichlet0 (inputParamRoutines.F90:3923)
    if (bc_coarse_corrections_enqueue(level + 1, 0.0)) return 1;
    launch_prolong_update_level(tf.tab, tc.tab, tc.n, tf.nx, tf.ny, tf.nz, kf, g_stream);
    const int secondHalo = (level <= g_opts.groundLevel);
    if (early_pressure_exchange_enqueue(level)) return 1;          // multiGrid.F90:602-604
    if (apply_bc_enqueue(level, secondHalo)) return 1;
    if (g_bc_callback) {
        HIPCHK(hipStreamSynchronize(g_stream));
        g_bc_callback(level, secondHalo);
    }
    return exchange_if_registered(level, secondHalo ? 2 : 1);
}

int adflow_gpu_transfer_to_coarse(int level)
{
    if (need_ready()) return 1;
    if (transfer_to_coarse_enqueue(level)) return 1;
    return sync_and_check();
}

int adflow_gpu_transfer_to_fine(int level)
{
    if (need_ready()) return 1;
    if (transfer_to_fine_enqueue(level)) return 1;
    return sync_and_check();
}

// coarseOwnedCoordinates(coarseLevel) (coarseUtils.F90:780-858) for every block pair (coarseLevel-1, coarseLevel)
int adflow_gpu_coarse_coordinates(int coarseLevel)
{
    if (need_ready()) return 1;
    if (coarseLevel < 2) return fail("coarse_coordinates: level %d has no finer level", coarseLevel);
    int rc = for_level_pairs(coarseLevel - 1, [&](Block* f, Block* c) {
        (void)f;
        if (!c->v.mgIFine) return fail("coarse block has no mgIFine maps");
        c->face_vectors_valid = false;
        return 0;
    });
    if (rc) return rc;
    LevelTab tf, tc;
    if (level_tab(coarseLevel - 1, &tf) || level_tab(coarseLevel, &tc)) return 1;
    launch_coarse_coordinates_level(tc.tab, tf.tab, tc.n, tc.nx, tc.ny, tc.nz, g_stream);
    return sync_and_check();
}

// executeMGCycle as ONE hipGraph.  A cycle enqueues ~10^3 kernels and most of them (the coarse levels) run 10 - 40 us: the launch
// gaps, not the kernels, bound them (profiles/r02_br_mg_trace.md).  Everything a cycle enqueues is fixed by the options, the tuning,
// the registered blocks / patterns / subfaces and the cycling strategy, so the third identical call in a row (the first warms the
// caches, no allocation may happen under capture; the second is captured) replays the graph.  Not captured: host callbacks
// (boundary-condition hooks), phase events, actuator regions (their relaxation factor changes from cycle to cycle), RCCL messages.
namespace {
struct MgGraph {
    std::vector<int32_t> cyc;
    long gen = -1;
    int seen = 0;                   // identical calls in a row that ran directly
    hipGraphExec_t exec = nullptr;
    hipGraph_t graph = nullptr;
    bool failed = false;            // capture or instantiation failed once: direct enqueue from then on
    // what is enqueued also depends on the blocks' host-side flags (a stale sensor adds a k_entropy pass, ...): the graph is valid
    // for the flags it was captured from (pre) and leaves the flags of the end of the captured cycle (post)
    std::vector<unsigned char> pre, post;
};
MgGraph g_mgg;

void mg_graph_drop()
{
    if (g_mgg.exec) (void)hipGraphExecDestroy(g_mgg.exec);
    if (g_mgg.graph) (void)hipGraphDestroy(g_mgg.graph);
    g_mgg.exec = nullptr; g_mgg.graph = nullptr; g_mgg.seen = 0; g_mgg.gen = -1; g_mgg.cyc.clear();
}

static std::vector<unsigned char> block_flags()
{
    std::vector<unsigned char> f;
    for (auto& kv : g_blocks) {
        const Block* b = kv.second;
        f.push_back((unsigned char)(b->geom_uploaded | (b->normals_from_x_ok << 1) | (b->face_vectors_valid << 2) | (b->ss_valid << 3) |
                                    (b->etot_consistent << 4)));
    }
    return f;
}

void set_block_flags(const std::vector<unsigned char>& f)
{
    size_t q = 0;
    for (auto& kv : g_blocks) {
        if (q >= f.size()) break;
        Block* b = kv.second;
        const unsigned char v = f[q++];
        b->geom_uploaded = v & 1; b->normals_from_x_ok = (v >> 1) & 1; b->face_vectors_valid = (v >> 2) & 1; b->ss_valid = (v >> 3) & 1;
        b->etot_consistent = (v >> 4) & 1;
    }
}

bool mg_graph_eligible()
{
#ifdef HOSTSIM
    return false;
#else
    if (!g_mg_graph || g_mgg.failed || g_bc_callback || g_turb_bc_callback || g_phase_base > 0 || !g_act.empty()) return false;
    for (auto& kv : g_comm)
        if (!kv.second.h_sendProc.empty() || !kv.second.h_recvProc.empty() || g_comm_self) return false;
    return true;
#endif
}
}  // namespace

static int mg_cycle_enqueue(const int32_t* cycling, int nSteps);

int adflow_gpu_mg_cycle(const int32_t* cycling, int nSteps)
{
    if (need_ready()) return 1;
    if (!cycling || nSteps < 1) return fail("mg_cycle: empty cycling strategy");
#ifndef HOSTSIM
    if (mg_graph_eligible()) {
        const bool same = (g_mgg.gen == g_state_gen && (int)g_mgg.cyc.size() == nSteps && !memcmp(g_mgg.cyc.data(), cycling, sizeof(int32_t) * nSteps));
        if (!same) {
            mg_graph_drop();
            g_mgg.cyc.assign(cycling, cycling + nSteps);
            g_mgg.gen = g_state_gen;
        }
        if (g_mgg.exec && block_flags() == g_mgg.pre) {
            HIPCHK(hipGraphLaunch(g_mgg.exec, g_stream));
            set_block_flags(g_mgg.post);
            return sync_and_check();
        }
        if (g_mgg.exec) {           // other flags than the captured cycle started from: this call runs directly
            g_mgg.seen = 0;
        } else if (g_mgg.seen >= 1) {
            // the previous identical call ran directly (every cache is warm): capture this one
            HIPCHK(hipStreamSynchronize(g_stream));
            g_mgg.pre = block_flags();
            if (hipStreamBeginCapture(g_stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const bool was = g_async;
                g_async = true;
                const int rc = mg_cycle_enqueue(cycling, nSteps);
                g_async = was;
                hipGraph_t gr = nullptr;
                const hipError_t e1 = hipStreamEndCapture(g_stream, &gr);
                if (rc == 0 && e1 == hipSuccess && gr && !(g_test_fault & 1) && hipGraphInstantiate(&g_mgg.exec, gr, nullptr, nullptr, 0) == hipSuccess) {
                    g_mgg.graph = gr;
                    g_mgg.post = block_flags();
                    HIPCHK(hipGraphLaunch(g_mgg.exec, g_stream));
                    return sync_and_check();
                }
                // capture failed: nothing was executed, but the enqueue walked the host-side block flags (ss_valid, etot_consistent,
                // face_vectors_valid) to their post-cycle values: put them back, then run the cycle directly -- for good, and also
                // when the enqueue itself reported an error under capture (the direct path reports it again if it is a real one)
                if (gr) (void)hipGraphDestroy(gr);
                (void)hipGetLastError();
                g_mgg.exec = nullptr;
                g_mgg.failed = true;
                set_block_flags(g_mgg.pre);
            } else {
                (void)hipGetLastError();
                g_mgg.failed = true;
            }
        }
        ++g_mgg.seen;
    }
#endif
    const bool was_async = g_async;
    g_async = true;                 // the whole cycle is enqueued, one sync at the end
    const int rc = mg_cycle_enqueue(cycling, nSteps);
    g_async = was_async;
    if (rc) return rc;
    return sync_and_check();
}

static int mg_cycle_enqueue(const int32_t* cycling, int nSteps)
{
    int level = g_opts.groundLevel;
    int rc = 0;
    for (int n = 0; n < nSteps && !rc; ++n) {
        switch (cycling[n]) {
        case -1:
            level -= 1;
            rc = transfer_to_fine_enqueue(level);
            break;
        case 0: {
            if (n > 0 && cycling[n - 1] != 1) {
                // time step + residual with rkStage = 0 (multiGrid.F90:880-888)
                int fwMode;
                const double rFil = rfil_stage0(&fwMode);
                KParams kp = make_kparams(level, rFil, fwMode);
                rc = time_step_level(level, kp);
                if (!rc) rc = enqueue_flow_residual(level, kp);
            }
            if (!rc) rc = (g_opts.smoother == ADFLOW_RUNGE_KUTTA) ? adflow_gpu_rk_smooth(level) : adflow_gpu_dadi_smooth(level);
            break;
        }
        case 1:
            rc = transfer_to_coarse_enqueue(level);
            level += 1;
            break;
        default:
            rc = fail("mg_cycle: cycling(%d) = %d is not -1, 0 or 1", n + 1, cycling[n]);
        }
    }
    if (!rc && g_opts.equations == ADFLOW_RANS) {
        // turbSolveDDADI on the ground level (multiGrid.F90:938)
        rc = adflow_gpu_sa_solve(g_opts.groundLevel);
    }
    if (!rc) {
        // closing time step + residual on the ground level (multiGrid.F90:944-950)
        level = g_opts.groundLevel;
        int fwMode;
        const double rFil = rfil_stage0(&fwMode);
        KParams kp = make_kparams(level, rFil, fwMode);
        rc = time_step_level(level, kp);
        if (!rc) rc = enqueue_flow_residual(level, kp);
    }
    return rc;
}

// --------------------------------------------------------- Newton-Krylov glue
namespace {
double* g_vec_dev = nullptr;     // device staging of the PETSc-layout vectors
size_t g_vec_elems = 0;

int vec_reserve(size_t n)
{
    if (n <= g_vec_elems) return 0;
    if (g_vec_dev) (void)hipFree(g_vec_dev);
    g_vec_dev = nullptr;
    g_vec_elems = 0;
    HIPCHK(hipMalloc((void**)&g_vec_dev, n * sizeof(double)));
    g_vec_elems = n;
    return 0;
}

long level1_dof(void)
{
    long n = 0;
    for (auto& kv : g_blocks)
        if (std::get<0>(kv.first) == 1) n += (long)kv.second->v.nx * kv.second->v.ny * kv.second->v.nz * kv.second->v.nw;
    return n;
}

// blocks in the reference's vector order: nn ascending (sps = 1)
int for_level1_in_order(const std::function<int(Block*, long)>& fn)
{
    std::map<int, Block*> byNN;
    for (auto& kv : g_blocks)
        if (std::get<0>(kv.first) == 1 && std::get<1>(kv.first) == 1) byNN[std::get<2>(kv.first)] = kv.second;
    if (byNN.empty()) return fail("no block registered on level 1");
    long off = 0;
    for (auto& kv : byNN) {
        int rc = fn(kv.second, off);
        if (rc) return rc;
        off += (long)kv.second->v.nx * kv.second->v.ny * kv.second->v.nz * kv.second->v.nw;
    }
    return 0;
}

int set_w_dev(const double* d_vec, bool withClosures = false)
{
    const double turbFloor = 1e-6 * g_opts.wInf[5];
    LevelTab t;
    if (level_tab(1, &t)) return 1;
    // block offsets: BlkView::vecOff; withClosures: the closures blocketteRes would start with, in the same pass
    if (withClosures) {
        KParams kp = make_kparams(1, 1.0, 0);
        if (!g_floor_flag_dev) HIPCHK(hipMalloc((void**)&g_floor_flag_dev, sizeof(int)));
        HIPCHK(hipMemsetAsync(g_floor_flag_dev, 0, sizeof(int), g_stream));
        launch_set_w_closures_level(t.tab, t.n, t.nx, t.ny, t.nz, d_vec, turbFloor, kp, g_floor_flag_dev, g_stream);
    } else
        launch_set_w_level(t.tab, t.n, t.nx, t.ny, t.nz, d_vec, turbFloor, g_stream);
    return for_level1_in_order([&](Block* b, long) {
        b->ss_valid = false;
        b->etot_consistent = false;
        return 0;
    });
}

int get_r_dev(double* d_vec, double turbScale, double* d_sums)
{
    LevelTab t;
    if (level_tab(1, &t)) return 1;
    launch_get_r_level(t.tab, t.n, t.nx, t.ny, t.nz, d_vec, turbScale, d_sums, g_stream);
    return 0;
}
}  // namespace

int adflow_gpu_set_w_vec(const double* wVec, long n)
{
    if (need_ready()) return 1;
    if (!wVec || n != level1_dof()) return fail("set_w_vec: n=%ld but the level-1 blocks hold %ld DOF", n, level1_dof());
    if (vec_reserve((size_t)n)) return 1;
    HIPCHK(hipMemcpyAsync(g_vec_dev, wVec, sizeof(double) * n, hipMemcpyHostToDevice, g_stream));
    if (set_w_dev(g_vec_dev)) return 1;
    HIPCHK(hipStreamSynchronize(g_stream));
    return 0;
}

static int get_vec_common(double* out, long n, double turbScale, double* sumsq2)
{
    if (need_ready()) return 1;
    if (!out || n != level1_dof()) return fail("get_r_vec: n=%ld but the level-1 blocks hold %ld DOF", n, level1_dof());
    if (vec_reserve((size_t)n)) return 1;
    if (!g_norm_dev) HIPCHK(hipMalloc((void**)&g_norm_dev, sizeof(double) * 8));
    HIPCHK(hipMemsetAsync(g_norm_dev, 0, sizeof(double) * 8, g_stream));
    if (get_r_dev(g_vec_dev, turbScale, sumsq2 ? g_norm_dev : nullptr)) return 1;
    HIPCHK(hipMemcpyAsync(out, g_vec_dev, sizeof(double) * n, hipMemcpyDeviceToHost, g_stream));
    if (sumsq2) HIPCHK(hipMemcpyAsync(sumsq2, g_norm_dev, sizeof(double) * 2, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    return 0;
}

int adflow_gpu_get_r_vec(double* rVec, long n, double* sumsq2) { return get_vec_common(rVec, n, g_opts.turbResScale, sumsq2); }
int adflow_gpu_get_res(double* res, long n) { return get_vec_common(res, n, 1.0, nullptr); }

static int nk_core_enqueue(bool closuresDone = false)
{
    // blocketteRes with its default arguments (blockette.F90:130-160): exact residual,
    // flow + turbulence, no intermediate update
    unsigned flags = (closuresDone ? 0u : ADFLOW_RES_CLOSURES) | ADFLOW_RES_HALO | ADFLOW_RES_FLOW;
    if (g_opts.equations == ADFLOW_RANS) flags |= ADFLOW_RES_TURB;
    return block_res_enqueue(1, flags);
}

// FormFunction_mf on device vectors (d_rVec may be d_wVec: every entry of w is consumed by setW before the first residual kernel starts)
static int nk_residual_enqueue(const double* d_wVec, double* d_rVec)
{
    if (set_w_dev(d_wVec, true)) return 1;
    g_etot_flag_level = 1;      // the owned energy is computeEtot(p) already wherever p kept its value (k_set_w_closures_level)
    // setRVec rides on the kernels that complete dw (Roe march, SA march) where those run; otherwise its own pass.  Actuator
    // sources are added to dw behind the core: then the vector is taken from dw afterwards.
    g_rvec_done = 0;
    g_rvec_target = g_act.empty() ? d_rVec : nullptr;
    const int rc = nk_core_enqueue(true);
    g_rvec_target = nullptr;
    g_etot_flag_level = 0;
    if (rc) return rc;
    const int need = (g_opts.equations == ADFLOW_RANS) ? 3 : 1;
    if ((g_rvec_done & need) != need)
        if (get_r_dev(d_rVec, g_opts.turbResScale, nullptr)) return 1;
    return 0;
}

int adflow_gpu_nk_residual_dev(const double* d_wVec, double* d_rVec, long n)
{
    if (need_ready()) return 1;
    if (!d_wVec || !d_rVec || n != level1_dof()) return fail("nk_residual: n=%ld but the level-1 blocks hold %ld DOF", n, level1_dof());
    if (nk_residual_enqueue(d_wVec, d_rVec)) return 1;
    return sync_and_check();
}

int adflow_gpu_nk_residual(const double* wVec, double* rVec, long n)
{
    if (need_ready()) return 1;
    if (!wVec || !rVec || n != level1_dof()) return fail("nk_residual: n=%ld but the level-1 blocks hold %ld DOF", n, level1_dof());
    if (vec_reserve((size_t)n)) return 1;
    HIPCHK(hipMemcpyAsync(g_vec_dev, wVec, sizeof(double) * n, hipMemcpyHostToDevice, g_stream));
    if (nk_residual_enqueue(g_vec_dev, g_vec_dev)) return 1;
    HIPCHK(hipMemcpyAsync(rVec, g_vec_dev, sizeof(double) * n, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    return 0;
}

// turbAPI::turbSolveDDADI (src/turbulence/turbAPI.F90:4-95) for Spalart-Allmaras:
// nSubIterTurb x [ sa_block(.false.) on every block ; whalo2(nt1:nt2, no p, viscosities) ]
int adflow_gpu_sa_solve(int level)
{
    if (need_ready()) return 1;
    if (g_opts.equations != ADFLOW_RANS) return fail("adflow_gpu_sa_solve needs equations = RANS");
    const int nit = std::max(1, (int)g_opts.nSubIterTurb);
    for (int it = 0; it < nit; ++it) {
        KParams kp = make_kparams(level, 1.0, 0);
        // sa_block(.false.) of every block: bcTurbTreatment first, applyAllTurbBCThisBlock(.true.) last (sa.F90:40-84)
        int rc = for_level(level, [&](Block* b) {
            if (b->v.nw < 6) return fail("RANS/SA needs nw = 6 (block has %d)", b->v.nw);
            return 0;
        });
        if (rc) return rc;
        if (turb_bc_treatment_enqueue(level, kp)) return 1;
        LevelTab t;
        if (level_tab(level, &t)) return 1;
        bool movingS = false;
        for_level(level, [&](Block* b) { movingS = movingS || b->v.sFace || b->v.moving; return 0; });
        const bool marchRes = (g_sa_march == 1 && !movingS);
        if (marchRes) {
            if (ensure_sa_tiles(level)) return 1;
            launch_sa_march(t.tab, g_sa_tiles[level].first, g_sa_tiles[level].second, kp, g_stream, true);
        }
        launch_sa_solve_level(t.tab, t.n, t.nx, t.ny, t.nz, kp, g_stream, marchRes);
        if (turb_bc_apply_enqueue(level, kp, 1)) return 1;
        if (g_turb_bc_callback) {
            HIPCHK(hipStreamSynchronize(g_stream));
            g_turb_bc_callback(level, 1);
        }
        if (g_comm.count(std::make_pair(level, 2)))
            if (halo_exchange_enqueue(level, 6, 6, 0, 1, 2)) return 1;
    }
    return sync_and_check();
}

int adflow_gpu_set_turb_bc_callback(adflow_bc_callback fn)
{
    ++g_state_gen;
    g_turb_bc_callback = fn;
    return 0;
}

int adflow_gpu_res_norms(int level, double* sums, int n)
{
    if (need_ready()) return 1;
    if (!sums || n < 1 || n > 6) return fail("res_norms: n must be 1..6");
    if (!g_norm_dev) HIPCHK(hipMalloc((void**)&g_norm_dev, sizeof(double) * 8));
    HIPCHK(hipMemsetAsync(g_norm_dev, 0, sizeof(double) * 8, g_stream));
    int rc = for_level(level, [&](Block* b) {
        if (n > b->v.nw) return fail("res_norms: n=%d exceeds nw=%d", n, b->v.nw);
        launch_res_norms(b->v, n, g_norm_dev, g_stream);
        return 0;
    });
    if (rc) return rc;
#ifndef ADFLOW_NO_RCCL
    if (g_nccl && g_nranks > 1) NCCLCHK(ncclAllReduce(g_norm_dev, g_norm_dev, n, ncclDouble, ncclSum, g_nccl, g_stream));
#endif
    HIPCHK(hipMemcpyAsync(sums, g_norm_dev, sizeof(double) * n, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    return 0;
}

// --------------------------------------------------------- instrumentation
int adflow_gpu_event_record(int slot)
{
    if (g_device < 0) return fail("adflow_gpu_init has not been called");
    if (slot < 0 || slot >= 64) return fail("event slot %d out of range", slot);
    HIPCHK(hipEventRecord(g_events[slot], g_stream));
    return 0;
}

int adflow_gpu_event_elapsed_ms(int a, int b, double* ms)
{
    if (a < 0 || a >= 64 || b < 0 || b >= 64 || !ms) return fail("bad event arguments");
    HIPCHK(hipEventSynchronize(g_events[b]));
    float f = 0.f;
    HIPCHK(hipEventElapsedTime(&f, g_events[a], g_events[b]));
    *ms = f;
    return 0;
}

int adflow_gpu_march_stats(int level, double* out, int n)
{
    if (need_ready()) return 1;
    if (!out || n < 4) return fail("march_stats: n = %d (>= 4)", n);
    for (int q = 0; q < n; ++q) out[q] = 0.0;
    if (ensure_gf_tiles(level) || ensure_sa_tiles(level) || ensure_tiles(level)) return 1;
    // Spalart-Allmaras march: one trip per produced plane, four wavefronts
    {
        const auto& tt = g_sa_tiles[level];
        std::vector<int4> h((size_t)tt.second);
        if (tt.second > 0) HIPCHK(hipMemcpy(h.data(), tt.first, sizeof(int4) * h.size(), hipMemcpyDeviceToHost));
        for (const int4& t : h)
            if (t.x >= 0) out[0] += 4.0 * (t.w - t.z + 1);
    }
    // fused gradient + viscous march: every chunk marches its planes + 2 (k_visc_gf: mm = k0-1 .. k1+1), four wavefronts
    {
        const auto& tt = g_gf_tiles[level];
        std::vector<int4> h((size_t)tt.second);
        if (tt.second > 0) HIPCHK(hipMemcpy(h.data(), tt.first, sizeof(int4) * h.size(), hipMemcpyDeviceToHost));
        for (const int4& t : h)
            if (t.x >= 0) out[1] += 4.0 * (t.w - t.z + 3);
    }
    for (auto& kv : g_blocks) {
        if (std::get<0>(kv.first) != level || std::get<1>(kv.first) != 1) continue;
        const BlkView& v = kv.second->v;
        // tile table (k_roe_march, k_visc_march, ...): 60 columns x march_by rows x march_kch planes, kch + 1 trips per chunk
        int ntx, nty, ntz;
        euler_march_tiles(v, &ntx, &nty, &ntz);
        out[2] += 4.0 * ntx * nty * (v.nz + ntz);
    }
    return 0;
}

int adflow_gpu_sync(void)
{
    if (g_device < 0) return fail("adflow_gpu_init has not been called");
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(g_stream));
    return 0;
}

int adflow_gpu_set_tuning(const char* key, int value)
{
    ++g_state_gen;
    if (!key) return fail("null tuning key");
    if (!strcmp(key, "euler_march")) { g_use_march = (value != 0); return 0; }
    if (!strcmp(key, "viscous_tiled")) { g_viscous_tiled = value; return 0; }
    if (!strcmp(key, "inviscid_march")) { g_inviscid_march = value; return 0; }
    if (!strcmp(key, "roe_march")) { g_roe_march = value; return 0; }
    if (!strcmp(key, "sa_march")) { g_sa_march = value; return 0; }
    if (!strcmp(key, "overlap")) { g_overlap = value; return 0; }
    if (!strcmp(key, "max_grid_z")) { g_max_grid_z = (value > 0) ? value : 65535; return 0; }
    if (!strcmp(key, "metric_from_x")) { g_metric_from_x = value; return 0; }
    if (!strcmp(key, "xcd_tiles")) {
        g_xcd_tiles = value;
        if (g_stream) (void)hipStreamSynchronize(g_stream);
        mg_graph_drop();       // a captured cycle holds the device pointer of the tile table (g_state_gen above retires it as well)
        for (auto& kv : g_tiles) (void)hipFree(kv.second.first);
        g_tiles.clear();
        return 0;
    }
    if (!strcmp(key, "phase_events")) {
        if (value != 0 && (value < 8 || value > 56)) return fail("phase_events: first slot must be 8..56 (or 0 = off)");
        g_phase_base = value;
        return 0;
    }
    if (!strcmp(key, "march_kch")) {
        if (value < 4) return fail("march_kch must be >= 4");
        g_march_kch = value;
        if (g_stream) (void)hipStreamSynchronize(g_stream);
        mg_graph_drop();
        for (auto* mp : {&g_tiles, &g_gf_tiles, &g_gf_tiles_int, &g_gf_tiles_bnd, &g_sa_tiles, &g_sa_tiles_int, &g_sa_tiles_bnd}) {
            for (auto& kv : *mp) (void)hipFree(kv.second.first);
            mp->clear();
        }
        return 0;
    }
    if (!strcmp(key, "split_eval")) { g_split_eval = value; return 0; }
    if (!strcmp(key, "test_fault")) { g_test_fault = value; g_mgg.failed = false; mg_graph_drop(); return 0; }
    if (!strcmp(key, "ad_cache")) { g_ad_cache = value; if (!value) { if (g_stream) (void)hipStreamSynchronize(g_stream); ad_drop(); } return 0; }
    if (!strcmp(key, "rvec_joint")) { g_rvec_joint = value; return 0; }
    if (!strcmp(key, "jac_snap")) { g_jac_snap = value; return 0; }
    if (!strcmp(key, "pc_fused")) { g_pc_fused = value; mg_graph_drop(); return 0; }
    if (!strcmp(key, "mg_graph")) { g_mg_graph = value; g_mgg.failed = false; mg_graph_drop(); return 0; }
    if (!strcmp(key, "comm_self")) {
        if (g_stream) (void)hipStreamSynchronize(g_stream);
        g_comm_self = value;
        for (auto& kv : g_comm) drop_comm_lists(kv.second);        // rebuilt at the next exchange
        return 0;
    }
    if (!strcmp(key, "dadi_pcr")) { g_dadi_pcr = value; mg_graph_drop(); return 0; }
    if (!strcmp(key, "ra_pcr")) { g_ra_pcr = value; mg_graph_drop(); return 0; }
    if (!strcmp(key, "dadi_upd")) { g_dadi_upd = value; mg_graph_drop(); return 0; }
    if (!strcmp(key, "gf_cus")) {
        // tests: the round size on a device with `value` CUs (0 = ask the device; -1 = chunks of march_kch planes, no fitting)
        g_gf_nofit = (value < 0);
        g_num_cus = value > 0 ? value : 0;
        if (g_stream) (void)hipStreamSynchronize(g_stream);
        mg_graph_drop();
        for (auto* mp : {&g_tiles, &g_gf_tiles, &g_gf_tiles_int, &g_gf_tiles_bnd, &g_sa_tiles, &g_sa_tiles_int, &g_sa_tiles_bnd}) {
            for (auto& kv : *mp) (void)hipFree(kv.second.first);
            mp->clear();
        }
        return 0;
    }
    return fail("unknown tuning key '%s'", key);
}

int adflow_gpu_set_async(int on)
{
    g_async = (on != 0);
    return 0;
}

int adflow_gpu_abi_sizes(int* opts_bytes, int* desc_bytes)
{
    if (opts_bytes) *opts_bytes = (int)sizeof(adflow_opts);
    if (desc_bytes) *desc_bytes = (int)sizeof(adflow_block_desc);
    return 0;
}

int adflow_gpu_abi_sizes2(int* bc_subface_bytes, int* comm_pattern_bytes)
{
    if (bc_subface_bytes) *bc_subface_bytes = (int)sizeof(adflow_bc_subface);
    if (comm_pattern_bytes) *comm_pattern_bytes = (int)sizeof(adflow_comm_pattern);
    return 0;
}

}  // extern "C"

// device buffers that outlive the blocks (vector staging, norm sums, surface nodes): released by adflow_gpu_finalize
static void free_side_buffers(void)
{
    free_xsurf();
    if (g_vec_dev) (void)hipFree(g_vec_dev);
    g_vec_dev = nullptr;
    g_vec_elems = 0;
    if (g_norm_dev) (void)hipFree(g_norm_dev);
    g_norm_dev = nullptr;
    if (g_floor_flag_dev) (void)hipFree(g_floor_flag_dev);
    g_floor_flag_dev = nullptr;
    if (g_snapreq.dev) (void)hipFree(g_snapreq.dev);
    g_snapreq = SnapReq();
#ifndef ADFLOW_NO_RCCL
    if (g_nccl) (void)ncclCommDestroy(g_nccl);
    g_nccl = nullptr;
    g_rank = 0;
    g_nranks = 1;
#endif
}
