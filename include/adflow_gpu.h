/*
 * adflow_gpu.h — C-ABI of the MI355X residual / smoother engine.
 *
 * This is the drop-in boundary of SURVEY.md §8(b).  The reference has no
 * plugin registry: its hot path is a set of Fortran module procedures that act
 * on module-global block pointers.  A maintainer keeps all host Fortran and
 * replaces the BODIES of the shell routines listed next to each entry point by
 * a call through an ISO_C_BINDING interface to the function below
 * (INTEGRATION.md shows the Fortran side; adflow_amd/fortran/adflow_gpu_shim.F90
 * is that interface module).
 *
 * Conventions
 *  - plain C types only; every function returns 0 on success, nonzero on error
 *    (adflow_gpu_last_error() gives the text; the Fortran shim forwards it to
 *    utils::terminate, src/utils/utils.F90:501).
 *  - host arrays are the reference's own Fortran arrays, column-major, with the
 *    bounds the reference allocates (SURVEY.md §8(a) row "T"); the library
 *    NEVER takes ownership — it only keeps device mirrors.
 *  - (nn, level, sps) identify a block exactly like flowDoms(nn,level,sps)
 *    (src/modules/block.F90:760-775); all three are 1-based.
 *  - reals are double (realType, src/modules/precision.F90:77-81), integers
 *    int32 (intType), porosities int8 (porType, precision.F90:110-111).
 *  - single host thread per process; all work is issued on one HIP stream per
 *    process and every entry point is synchronous on return unless stated.
 */
#ifndef ADFLOW_GPU_H
#define ADFLOW_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enumerations: values of src/modules/constants.F90 */
enum { ADFLOW_EULER = 1, ADFLOW_NS = 2, ADFLOW_RANS = 3 };
enum { ADFLOW_DISS_SCALAR = 1, ADFLOW_DISS_MATRIX = 2, ADFLOW_UPWIND = 9 };
enum { ADFLOW_LIM_FIRST_ORDER = 1, ADFLOW_LIM_NONE = 2, ADFLOW_LIM_VANALBADA = 3, ADFLOW_LIM_MINMOD = 4 };
enum { ADFLOW_RUNGE_KUTTA = 1, ADFLOW_DADI = 2 };
enum { ADFLOW_TURBPROD_STRAIN = 1, ADFLOW_TURBPROD_VORTICITY = 2 };
enum { ADFLOW_RESAVG_NEVER = 0, ADFLOW_RESAVG_ALWAYS = 1, ADFLOW_RESAVG_ALTERNATE = 2 };

#define ADFLOW_MAX_RK_STAGES 8

/* host hook type: see adflow_gpu_set_bc_callback */
typedef void (*adflow_bc_callback)(int level, int secondHalo);

/* boundary subfaces: BCType and BCFaceID values of src/modules/constants.F90:257-297 */
enum {
    ADFLOW_BC_SYMM = -1, ADFLOW_BC_SYMM_POLAR = -2, ADFLOW_BC_NSWALL_ADIABATIC = -3, ADFLOW_BC_NSWALL_ISOTHERMAL = -4,
    ADFLOW_BC_EULERWALL = -5, ADFLOW_BC_FARFIELD = -6, ADFLOW_BC_SUPERSONIC_INFLOW = -7, ADFLOW_BC_SUBSONIC_INFLOW = -8,
    ADFLOW_BC_SUPERSONIC_OUTFLOW = -9, ADFLOW_BC_SUBSONIC_OUTFLOW = -10, ADFLOW_BC_MASSBLEED_OUTFLOW = -12,
    ADFLOW_BC_EXTRAP = -15
};
enum { ADFLOW_INLET_TOTAL_CONDITIONS = 1, ADFLOW_INLET_MASS_FLOW = 2 };   /* BCData%subsonicInletTreatment, constants.F90:237 */
enum { ADFLOW_IMIN = 1, ADFLOW_IMAX = 2, ADFLOW_JMIN = 3, ADFLOW_JMAX = 4, ADFLOW_KMIN = 5, ADFLOW_KMAX = 6 };
enum { ADFLOW_WALLBC_CONSTANT = 1, ADFLOW_WALLBC_LINEAR = 2, ADFLOW_WALLBC_QUADRATIC = 3, ADFLOW_WALLBC_NORMAL_MOMENTUM = 4 };

/* One boundary subface of a block: flowDoms(nn,level,sps)%BCType(mm), %BCFaceID(mm) and the
 * members of %BCData(mm) the flow boundary conditions read (block.F90 BCDataType).  Every array
 * has the bounds (icBeg:icEnd, jcBeg:jcEnd [,3]) of the reference, Fortran order; NULL where the
 * boundary type does not use it (rface NULL = zero grid velocity). */
typedef struct adflow_bc_subface {
    int32_t bcType, faceID;
    int32_t icBeg, icEnd, jcBeg, jcEnd;
    int32_t subsonicInletTreatment;   /* SubsonicInflow: 1 total conditions, 2 mass flow                 */
    int32_t reserved;
    const double* norm;       /* unit outward normal, 3 components                                       */
    const double* rface;      /* normal grid velocity (EulerWall, farField)                              */
    const double* uSlip;      /* wall velocity, 3 components (NSWall*)                                   */
    const double* TNS_Wall;   /* wall temperature (NSWallIsothermal)                                     */
    const double* rho;        /* prescribed state (SupersonicInflow; SubsonicInflow with mass flow): rho, velx, vely, velz */
    const double* velx;
    const double* vely;
    const double* velz;
    const double* ps;         /* static pressure (SupersonicInflow, SubsonicOutflow, MassBleedOutflow)   */
    const double* ptInlet;    /* SubsonicInflow with total conditions: total pressure, temperature, enthalpy, */
    const double* ttInlet;    /* unit flow direction                                                     */
    const double* htInlet;
    const double* flowXdirInlet;
    const double* flowYdirInlet;
    const double* flowZdirInlet;
    const double* turbInlet;  /* prescribed turbulence variable(s) of inflow subfaces (RANS), (:,:,nt1:nt2) */
    double symNorm[3];        /* BCData%symNorm: the (constant) normal of a symmetry plane, read by xhalo_block */
} adflow_bc_subface;

/* Options: snapshot of the Fortran module variables the hot path reads.
 * Refreshed by the shim at every entry (Python may assign them between calls,
 * adflow/pyADflow.py:5463-5630).  Field names are the reference's. */
typedef struct adflow_opts {
    /* inputPhysics (src/modules/inputParam.F90:507-635) */
    int32_t equations, turbModel, turbProd;
    int32_t useQCR, useRotationSA, useft2SA;
    /* inputDiscretization (inputParam.F90:1-97) */
    int32_t spaceDiscr, spaceDiscrCoarse, limiter, orderTurb;
    int32_t dirScaling;
    /* inputIteration (inputParam.F90:183-299) */
    int32_t smoother, nRKStages, resAveraging, nSubiterations, nSubIterTurb;
    /* iteration (src/modules/iteration.f90) */
    int32_t groundLevel;
    int32_t turbRelax;        /* inputIteration: 1 explicit, 2 implicit (default for SA, inputParamRoutines.F90:3402) */
    /* inputDiscretization: boundary treatment (constants.F90:170-178): 1 constant, 2 linear, 4 normal momentum;
     * outflowTreatment 1 constant, 2 linear extrapolation */
    int32_t eulerWallBCTreatment, viscWallBCTreatment, outflowTreatment;
    int32_t hScalingInlet;            /* inputDiscretization: total-enthalpy scaling of the subsonic-inflow Riemann invariant */
    /* features of the reference this library does NOT implement, as a bit mask the host fills from its option modules:
     * 1 equationMode /= steady (unsteady / time spectral), 2 cpModel /= cpConstant, 4 wall functions, 8 overset blocks present.
     * adflow_gpu_set_options refuses a non-zero mask instead of silently returning steady, constant-gamma, 1-to-1 results. */
    int32_t unsupported;
    int32_t lowSpeedPreconditioner;   /* inputDiscretization: residual_block's 5x5 low-Mach transform (residuals.F90:172-331) + the 0.8 RK step factor (smoothers.F90:202) */
    /* iteration::exchangePressureEarly (iteration.f90:44-53; set by solvers.F90:35-39,140-144 = eulerWallBcTreatment == normalMomentum
     * .and. EulerWallsPresent(), a reduction over ALL processes): pressure-only whalo1 before applyAllBC in every smoother stage
     * and in transferToFineGrid (smoothers.F90:363,674, multiGrid.F90:602) */
    int32_t exchangePressureEarly;
    int32_t reserved_i;
    double gammaConstant, prandtl, prandtlTurb;
    double SSuthDim, muSuthDim, TSuthDim;
    double SAKappa, SAcb1, SAcb2, SAsigma, SAcv1, SAcw1, SAcw2, SAcw3, SAct1, SAct2, SAct3, SAct4, SAcrot;
    double vis2, vis4, vis2Coarse, adis, acousticScaleFactor, kappaCoef;
    double cfl, cflCoarse, cflLimit, fcoll, smoop, alfaTurb, betaTurb, turbResScale;
    double etaRK[ADFLOW_MAX_RK_STAGES], cdisRK[ADFLOW_MAX_RK_STAGES];
    /* flowVarRefState (src/modules/flowVarRefState.F90) */
    double gammaInf, pInf, pInfCorr, rhoInf, uInf, RGas, muInf, muRef, TRef, timeRef;
    double wInf[10];
    double sigma;             /* inputDiscretization: lumped-dissipation coefficient of the approximate residual */
    double pRef, uRef, LRef;  /* flowVarRefState: scales of the actuator-region source terms (residuals.F90:370-385) */
    double ordersConverged;   /* iteration: relaxation of the actuator source between relaxStart and relaxEnd */
    double reserved_d[3];
} adflow_opts;

/* Host arrays of one block, flowDoms(nn,level,sps)%... .  NULL = not present
 * (e.g. rev for Euler).  Bounds in comments are the reference's allocation. */
typedef struct adflow_block_desc {
    int32_t nx, ny, nz;       /* owned cells: il=nx+1, ie=nx+2, ib=nx+3 (block.F90:363-390) */
    int32_t nw;               /* 5, or 6 with Spalart-Allmaras */
    int32_t rightHanded;      /* blockType%rightHanded */
    int32_t reserved;
    /* state (initializeFlow.F90:457-529) */
    double *w;                /* (0:ib,0:jb,0:kb,1:nw)  rho,u,v,w,rhoE[,nuTilde] */
    double *p, *gamma;        /* (0:ib,0:jb,0:kb) */
    double *rlv, *rev;        /* (0:ib,0:jb,0:kb) */
    /* geometry */
    double *x;                /* (0:ie,0:je,0:ke,3)  partitioning.F90:1761 */
    double *sI, *sJ, *sK;     /* (0:ie,1:je,1:ke,3) (1:ie,0:je,1:ke,3) (1:ie,1:je,0:ke,3) */
    double *vol, *volRef;     /* (0:ib,0:jb,0:kb) */
    double *d2Wall;           /* (2:il,2:jl,2:kl)  wallDistance.F90:503 */
    int8_t *porI, *porJ, *porK; /* (1:il,2:jl,2:kl) (2:il,1:jl,2:kl) (2:il,2:jl,1:kl) preprocessingAPI.F90:567 */
    int32_t *iblank;          /* (0:ib,0:jb,0:kb) */
    /* residual / work arrays the host may want back (utils.F90:3969-3974) */
    double *dw;               /* (0:ib,0:jb,0:kb,1:nw) */
    double *fw;               /* (0:ib,0:jb,0:kb,1:5) */
    double *dtl, *radI, *radJ, *radK; /* (1:ie,1:je,1:ke) */
    /* multigrid work (coarse levels, initializeFlow.F90:746-748) */
    double *w1, *p1, *wr;     /* (1:ie,1:je,1:ke,1:5) (1:ie,1:je,1:ke) (2:il,2:jl,2:kl,1:5) */
    /* multigrid transfer maps (src/preprocessing/coarseUtils.F90:254-262), NULL when absent.
     * On a COARSE block: the two fine cells of each coarse cell and the restriction weights */
    int32_t *mgIFine, *mgJFine, *mgKFine;       /* (1:ie,2) (1:je,2) (1:ke,2) */
    double *mgIWeight, *mgJWeight, *mgKWeight;  /* (2:il) (2:jl) (2:kl) */
    /* on a FINE block: nearest and next-nearest coarse cell of each fine cell */
    int32_t *mgICoarse, *mgJCoarse, *mgKCoarse; /* (2:il,2) (2:jl,2) (2:kl,2) */
    /* moving blocks (rotating frame / ALE, block.F90 sFaceI/J/K, addGridVelocities, blockIsMoving): dot product of the
     * face velocity with the face normal; NULL / 0 for a block at rest.  rotRate = cgnsDoms(nbkGlobal)%rotRate, used by
     * the rotational source of inviscidCentralFlux (fluxes.F90:372-397) when blockIsMoving in steady mode */
    double *sFaceI, *sFaceJ, *sFaceK;           /* (0:ie,1:je,1:ke) (1:ie,0:je,1:ke) (1:ie,1:je,0:ke) */
    double rotRate[3];
    int32_t addGridVelocities, blockIsMoving;
} adflow_block_desc;

/* 1-to-1 halo communication pattern of one level and one halo depth: the
 * reference's internalCell_{1st,2nd}(level) and commPatternCell_{1st,2nd}(level)
 * (src/modules/communication.F90 internalCommType / commType), flattened.  Cell
 * indices are the values the reference stores (actual cell indices 0..ib; its
 * "+1" at use, haloExchange.F90:605-607, only undoes a pointer rebasing).
 * Index arrays are column-major (n,3) like the Fortran ones; block ids are the
 * local block numbers nn (1-based).  Entry order inside a message must match on
 * sender and receiver, which the reference's preprocessing guarantees. */
typedef struct adflow_comm_pattern {
    int32_t ncopy;                       /* internal%ncopy: same-process copies */
    const int32_t *donorBlock, *donorIndices;   /* (ncopy), (ncopy,3) */
    const int32_t *haloBlock, *haloIndices;     /* (ncopy), (ncopy,3) */
    int32_t nProcSend;                   /* commPattern%nProcSend */
    const int32_t *sendProc;             /* (nProcSend) ranks */
    const int32_t *nsendCum;             /* (0:nProcSend) cumulative cell counts, nsendCum[0] = 0 */
    const int32_t *sendBlock, *sendIndices;     /* (nsendCum[nProcSend]), (..,3): sendList(i)%block / %indices concatenated */
    int32_t nProcRecv;
    const int32_t *recvProc;
    const int32_t *nrecvCum;
    const int32_t *recvBlock, *recvIndices;
} adflow_comm_pattern;

/* identifiers for adflow_gpu_download_array / adflow_gpu_upload_array */
enum {
    ADFLOW_ARR_W = 1, ADFLOW_ARR_P, ADFLOW_ARR_GAMMA, ADFLOW_ARR_RLV, ADFLOW_ARR_REV,
    ADFLOW_ARR_DW, ADFLOW_ARR_FW, ADFLOW_ARR_DTL, ADFLOW_ARR_RADI, ADFLOW_ARR_RADJ, ADFLOW_ARR_RADK,
    ADFLOW_ARR_AA, ADFLOW_ARR_NODAL_GRADS, ADFLOW_ARR_WN, ADFLOW_ARR_PN, ADFLOW_ARR_W1, ADFLOW_ARR_P1,
    ADFLOW_ARR_WR, ADFLOW_ARR_VOL, ADFLOW_ARR_SI, ADFLOW_ARR_SJ, ADFLOW_ARR_SK, ADFLOW_ARR_X,
    ADFLOW_ARR_D2WALL                  /* d2Wall(2:il,2:jl,2:kl) */
};

/* flags of adflow_gpu_block_res: the logical arguments of blockette::blocketteRes
 * (src/NKSolver/blockette.F90:70-120) */
enum {
    ADFLOW_RES_UPDATE_INTERMED = 1u,   /* also store dtl, radI/J/K.  WITHOUT it the spectral radii and dtl are not outputs of the call
                                          (as in blocketteResCore, which keeps them in tile-private arrays): for matrix dissipation
                                          and Roe upwind they are not formed at all, and ADFLOW_ARR_RADI/J/K, ADFLOW_ARR_DTL on the
                                          device are UNDEFINED afterwards (adflow_gpu_time_step or a call with this flag refreshes
                                          them; the smoothers call the time step themselves) */
    ADFLOW_RES_FLOW = 2u,              /* useFlowRes  */
    ADFLOW_RES_TURB = 4u,              /* useTurbRes  */
    /* the part of blocketteRes in front of the core (blockette.F90:195-246): */
    ADFLOW_RES_CLOSURES = 8u,          /* computePressureSimple + laminar/eddy viscosity, owned cells */
    ADFLOW_RES_HALO = 16u,             /* boundary-condition hook + whalo2(1, lStart, lEnd, T,T,T) */
    /* approximate residual of the preconditioner assembly (blockette.F90:755-852, fluxes.F90:3487-4975): */
    ADFLOW_RES_DISS_APPROX = 32u,      /* useDissApprox: inviscidDissFluxScalarApprox / MatrixApprox (lumped 2nd-difference
                                          dissipation with the FROZEN sensor of adflow_gpu_reference_shock_sensor) */
    ADFLOW_RES_VISC_APPROX = 64u,      /* useViscApprox: viscousFluxApprox (thin-layer normal differences) */
    /* with DISS_APPROX and the Roe upwind scheme the two residual cores of the reference differ: blocketteResCore (useBlockettes = T,
     * the default) calls inviscidUpwindFlux(.False.) = first-order reconstruction (blockette.F90:643), blockResCore keeps the
     * limiter (:827).  The host passes this flag when inputDiscretization::useBlockettes is set. */
    ADFLOW_RES_UPWIND_FIRST_ORDER = 128u
};

/* ---- lifetime ---------------------------------------------------------- */
int adflow_gpu_init(int device_ordinal);
int adflow_gpu_finalize(void);
const char* adflow_gpu_last_error(void);
int adflow_gpu_device_name(char* buf, int len);

/* ---- multi-GPU (RCCL over xGMI) replaces the MPI path of
 *      src/utils/haloExchange.F90:553-719 ------------------------------------ */
int adflow_gpu_comm_unique_id(void* id128);                       /* rank 0: create id (128 bytes) */
int adflow_gpu_comm_init(int rank, int nranks, const void* id128); /* all ranks */
/* What the communicator itself reports (ncclCommCount / ncclCommUserRank; -1 / -1 before adflow_gpu_comm_init) beside the rank and
 * size the library was given: lets a launcher check that every rank joined ONE communicator of the expected size before the first
 * exchange (the reference's myID / nProc of communication.F90 come from MPI_Comm_rank / _size the same way). */
int adflow_gpu_comm_info(int* rank, int* nranks, int* commCount, int* commUserRank);

/* ---- data model -------------------------------------------------------- */
int adflow_gpu_block_register(int nn, int level, int sps, const adflow_block_desc* d);
/* free the device mirrors of one block / of every block (utils::releaseMemoryPart1/2,
 * src/utils/utils.F90:4253,4732); host arrays are untouched */
int adflow_gpu_block_release(int nn, int level, int sps);
int adflow_gpu_release_all(void);
int adflow_gpu_upload_geometry(int nn, int level, int sps);   /* x,sI,sJ,sK,vol,volRef,d2Wall,por*,iblank */
/* Mesh warping ("next" row 3 of SURVEY.md 8f): upload the node coordinates only, then derive cell volumes, face
 * normals and the unit normals of the registered boundary subfaces on the device = volume_block + metric_block +
 * boundaryNormals (src/adjoint/adjointExtra.F90:5-364), the `useSpatial` branch of blocketteRes (blockette.F90:203-211) */
int adflow_gpu_upload_coordinates(int nn, int level, int sps);
int adflow_gpu_update_geometry(int level);
/* wallDistance::updateWallDistancesQuickly (src/wallDistance/wallDistance.F90:36-120), called by the `useSpatial` branch for RANS
 * with useApproxWallDistance (blockette.F90:207-209): d2Wall of the owned cells = | cell centre - closest wall point |, the wall
 * point re-formed from the association the host found once (determineWallAssociation, :1663-2002) and the current surface
 * coordinates.  register: flowDoms(nn,level,sps)%surfNodeIndices(4,2:il,2:jl,2:kl) (1-based numbers into xSurf, first = 0: no wall
 * within reach, d2Wall = large) and %uv(2,2:il,2:jl,2:kl).  update: xSurf = the scattered surface-node vector of updateXSurf
 * (:2004-2051), n = 3 x nodes; runs after adflow_gpu_update_geometry's coordinates are in place */
int adflow_gpu_wall_distance_register(int nn, int level, int sps, const int32_t* surfNodeIndices, const double* uv);
int adflow_gpu_update_wall_distances(int level, const double* xSurf, int64_t n);
/* Halo node coordinates after the owned nodes moved, the two steps that precede volume / metric in the `useSpatial`
 * branch of blocketteRes (blockette.F90:181-187):
 *   adflow_gpu_xhalo          adjointExtra::xhalo_block (adjointExtra.F90:365-599) of every block of the level: linear
 *                             extrapolation of the halo nodes 0 / ie, je, ke, then the mirror image in symmetry planes
 *                             (subfaces of kind symm with their BCData%symNorm);
 *   adflow_gpu_exchange_coor  haloExchange::exchangeCoor (haloExchange.F90:2456-2640): halo nodes of 1-to-1 interfaces
 *                             from the neighbours' interior nodes, using the NODE pattern commPatternNode_1st(level) /
 *                             internalNode_1st(level) registered with adflow_gpu_comm_register(level, 0, pattern)
 *                             (same-GPU copies + RCCL send/recv; periodic transformations as registered with
 *                             adflow_gpu_comm_register_periodic). */
int adflow_gpu_xhalo(int level);
/* coarseUtils::coarseOwnedCoordinates(coarseLevel) (coarseUtils.F90:780-858): the owned nodes of the coarse blocks from
 * the level above, through the registered mgI/J/KFine maps.  updateCoordinatesAllLevels / updateMetricsAllLevels
 * (preprocessingAPI.F90:3945-4017) on the device = for every coarse level: this, adflow_gpu_xhalo,
 * adflow_gpu_exchange_coor, adflow_gpu_update_geometry. */
int adflow_gpu_coarse_coordinates(int coarseLevel);
int adflow_gpu_exchange_coor(int level);
int adflow_gpu_upload_state(int nn, int level, int sps);      /* w,p,gamma,rlv,rev incl. both halo layers */
int adflow_gpu_download_state(int nn, int level, int sps);
int adflow_gpu_download_residual(int nn, int level, int sps); /* dw -> desc.dw */
int adflow_gpu_download_array(int nn, int level, int sps, int which, double* host);
int adflow_gpu_upload_array(int nn, int level, int sps, int which, const double* host);
int adflow_gpu_set_options(const adflow_opts* o);

/* ---- the hot path; each acts on ALL registered blocks of `level`, like the
 *      reference's shell loops over nDom -------------------------------------- */
/* solverUtils::timeStep (src/solver/solverUtils.F90:4-41, block body :43-356) */
int adflow_gpu_time_step(int level, int onlyRadii);
/* residuals::initres (src/solver/residuals.F90:964-1026, block body :427-955), 1-based var range */
int adflow_gpu_initres(int level, int varStart, int varEnd);
/* residuals::residual (src/solver/residuals.F90:1028-1060, block body :4-346);
 * rkStage selects rFil = cdisRK(rkStage+1) for the Runge-Kutta smoother */
int adflow_gpu_residual(int level, int rkStage);
/* blockette::blocketteRes main loop (src/NKSolver/blockette.F90:266-283):
 * timeStep + initres + [SA] + inviscid + [viscous] + dw=(dw+fw)*iblank, rFil=1 */
int adflow_gpu_block_res(int level, unsigned flags);
/* adjointUtils::referenceShockSensor (src/adjoint/adjointUtils.F90:1909-1969): freeze the shock sensor of every
 * level-1 block at the current state (pressure for Euler / matrix dissipation, entropy p/rho^gamma otherwise) for
 * the following ADFLOW_RES_DISS_APPROX evaluations */
int adflow_gpu_reference_shock_sensor(int level);
/* smoothers::RungeKuttaSmoother / DADISmoother (src/solver/smoothers.F90:4,383) */
int adflow_gpu_rk_smooth(int level);
int adflow_gpu_dadi_smooth(int level);
/* turbAPI::turbSolveDDADI for Spalart-Allmaras (src/turbulence/turbAPI.F90:4-95, sa.F90:16-86,717-1268):
 * nSubIterTurb x [SA residual + central jacobian, DDADI line solves j,i,k, update of
 * nuTilde and rev, turbulent-BC hook, whalo2(nt1:nt2)] */
int adflow_gpu_sa_solve(int level);
int adflow_gpu_set_turb_bc_callback(adflow_bc_callback fn);   /* applyAllTurbBCThisBlock stays on the host */
/* multigrid::transferToCoarseGrid (src/solver/multiGrid.F90:5-324): residual on `level`,
 * volume-weighted restriction to level+1, coarse residual, forcing term wr */
int adflow_gpu_transfer_to_coarse(int level);
/* multigrid::transferToFineGrid(.true.) (multiGrid.F90:326-652): prolongation of the
 * corrections of level+1 to `level`, state update, halo exchange */
int adflow_gpu_transfer_to_fine(int level);
/* multigrid::executeMGCycle (multiGrid.F90:825-955): `cycling` as produced by
 * setCycleStrategy (:957-1030): -1 prolongate, 0 smooth, +1 restrict; ends with the
 * turbSolveDDADI (RANS) and the ground-level time step + residual. */
int adflow_gpu_mg_cycle(const int32_t* cycling, int nStepsCycling);
/* register the 1-to-1 pattern of (level, nLayers = 1 | 2); lists are copied */
int adflow_gpu_comm_register(int level, int nLayers, const adflow_comm_pattern* p);
/* Periodic transformations of a registered pattern: the periodicData(:) of internalCell_*(level) AND commPatternCell_*(level)
 * (communication.F90 periodicDataType) concatenated - they address disjoint halos.  Applied on the receiving side after the
 * exchange: velocities of the listed halo cells rotated by rotMatrix when the exchanged range covers ivx..ivz
 * (correctPeriodicVelocity, haloExchange.F90:456-551); for the node pattern (nLayers = 0) the halo node coordinates become
 * rotMatrix (x - rotCenter) + translation + rotCenter (correctPeriodicCoor, haloExchange.F90:2644-2712).
 * Call after adflow_gpu_comm_register of the same (level, nLayers); nPeriodic = 0 removes them. */
typedef struct adflow_periodic_data {
    double rotMatrix[9];            /* (3,3) column-major, as stored */
    double rotCenter[3], translation[3];
    int32_t nHalos, reserved;
    const int32_t *block, *indices; /* (nHalos), (nHalos,3) column-major */
} adflow_periodic_data;
int adflow_gpu_comm_register_periodic(int level, int nLayers, int nPeriodic, const adflow_periodic_data* pd);
/* haloExchange::whalo1 (nLayers=1) / whalo2 (nLayers=2) (src/utils/haloExchange.F90:5,109):
 * w(varStart:varEnd) [+ p] [+ rlv, rev when viscous / eddy model]; 1-based variable range;
 * same-process copies on the device, other ranks through RCCL send/recv */
int adflow_gpu_halo_exchange(int level, int varStart, int varEnd, int commPressure, int commVisc, int nLayers);
/* split form of the inter-process part, for a caller-owned transport (used by the
 * CPU multi-process tests; the product path is adflow_gpu_halo_exchange):
 * pack the message for send slot `islot` (0-based) into `buf` (host or device
 * memory of nvar*count doubles, variable-major); unpack recv slot `islot` from `buf`;
 * *_count return the number of cells of the slot and the peer rank */
int adflow_gpu_halo_slot_info(int level, int nLayers, int isSend, int islot, int* peer, int* count);
int adflow_gpu_halo_pack(int level, int nLayers, int islot, int varStart, int varEnd, int commPressure, int commVisc, double* buf);
int adflow_gpu_halo_unpack(int level, int nLayers, int islot, int varStart, int varEnd, int commPressure, int commVisc, const double* buf);
int adflow_gpu_halo_local_copy(int level, int nLayers, int varStart, int varEnd, int commPressure, int commVisc);
/* host hook called between the state update and the halo exchange of every
 * smoother stage, where the reference applies boundary conditions
 * (applyAllBC, smoothers.F90:369,680).  NULL (default) = no physical boundaries. */
int adflow_gpu_set_bc_callback(adflow_bc_callback fn);
/* Boundary conditions on the device ("next" row 1 of SURVEY.md §8f).
 * adflow_gpu_bc_register copies the subfaces of one block (the first nViscBocos are the viscous
 * walls, as in the reference) to the device; adflow_gpu_apply_all_bc is BCRoutines::applyAllBC
 * (src/solver/BCRoutines.F90:15-221: symm -> adiabatic wall -> isothermal wall -> farfield ->
 * extrap / supersonic outflow -> Euler wall -> supersonic inflow, subfaces in index order inside
 * each kind, computeEtot + extrapolate2ndHalo as there) for every registered block of the level.
 * Once a level has registered subfaces the smoothers, the multigrid transfers and the NK residual
 * apply them on the device at the points where the reference calls applyAllBC; the host callback
 * (if any) still runs afterwards for kinds that are not implemented here (bleed inflow, mDot / thrust,
 * domain interfaces, sliding interfaces; normal-momentum Euler wall: registration of those returns an error). */
int adflow_gpu_bc_register(int nn, int level, int sps, int nBocos, int nViscBocos, const adflow_bc_subface* faces);
int adflow_gpu_apply_all_bc(int level, int secondHalo);
/* viscSubface(mm)%tau(:,:,1:6) and %q(:,:,1:3) of viscous subface mm (1-based, mm <= nViscBocos): the wall stress tensor and
 * heat flux that viscousFlux stores when rkStage == 0 on the ground level (storeWallTensor, fluxes.F90:2586-2592, 2861-2892)
 * and that the host's force integration reads (surfaceIntegrations.F90:718).  The arrays cover the owned face cells
 * inBeg+1:inEnd x jnBeg+1:jnEnd as allocated by viscSubfaceInfo (preprocessingAPI.F90:2520-2541); either may be NULL.
 * The device stores them after every such residual evaluation (adflow_gpu_residual with rkStage 0, the D-ADI smoother,
 * the multigrid cycle's closing residual, adflow_gpu_block_res). */
int adflow_gpu_download_wall_stress(int nn, int level, int sps, int mm, double* tau, double* q);
/* Actuator regions (actuatorRegionData.F90): residuals::sourceTerms_block (residuals.F90:348-425) adds the body force and
 * heat source of every listed cell to dw of the FINE level: -vol * force / volume / pRef on the momentum residuals,
 * -(F . v) - vol * heat / volume / (pRef uRef LRef^2) on the energy residual, ramped by ordersConverged between
 * relaxStart and relaxEnd.  Once regions are registered every level-1 residual evaluation of the library includes
 * them where the reference calls sourceTerms (smoothers.F90:74,409, multiGrid.F90:52,887,949, blockette.F90:278);
 * the body of the host's `sourceTerms` shell becomes a no-op like `initres`.  block(:) = local block nn of each cell
 * (what blkPtr encodes), cellIDs (3,nCellIDs) column-major.  nRegions = 0 removes them. */
typedef struct adflow_actuator_region {
    int32_t nCellIDs, reserved;
    const int32_t *block, *cellIDs;
    double force[3], heat, volume, relaxStart, relaxEnd;
} adflow_actuator_region;
int adflow_gpu_actuator_register(int nRegions, const adflow_actuator_region* regions);
/* sum over owned cells of (dw(:,l)/vol)^2, l=1..n  (solvers.F90:1538) */
int adflow_gpu_res_norms(int level, double* sums, int n);

/* ---- Newton-Krylov glue (src/NKSolver/NKSolvers.F90) -----------------------
 * vectors are the PETSc layout: block, k, j, i, variable fastest; n = total DOF of
 * the level-1 blocks of this process (nw * owned cells).  Host-pointer forms copy
 * through PCIe; *_dev forms take device pointers (PETSc VECHIP). */
int adflow_gpu_set_w_vec(const double* wVec, long n);                 /* setW :1331 (turbulence clipped at 1e-6*wInf) */
int adflow_gpu_get_r_vec(double* rVec, long n, double* sumsq2);        /* setRVec :1262: dw/volRef, turb*turbResScale; sumsq2[0..1] = sum flow^2, turb^2 (may be NULL) */
int adflow_gpu_get_res(double* res, long n);                           /* nksolver::getRes :1413: no turbResScale */
/* FormFunction_mf (:437-461): setW + blocketteRes(all defaults) + setRVec */
int adflow_gpu_nk_residual(const double* wVec, double* rVec, long n);
int adflow_gpu_nk_residual_dev(const double* d_wVec, double* d_rVec, long n);

/* ---- instrumentation: HIP events on the library's own stream ------------ */
int adflow_gpu_event_record(int slot);                   /* slot in [0,64) */
int adflow_gpu_event_elapsed_ms(int slot_start, int slot_stop, double* ms);
int adflow_gpu_sync(void);
/* wavefront-steps (wavefronts x k-planes marched) one evaluation of `level` costs each marching kernel, for the FP64-issue roofline of
 * bench.py (steps x the instruction count of the kernel's main loop, profiles/isa_counts.json): out[0] Spalart-Allmaras march,
 * [1] fused nodal gradients + viscous fluxes, [2] inviscid / viscous marches over the tile table, [3] reserved (0);
 * n >= 4.  Geometry only, no device work. */
int adflow_gpu_march_stats(int level, double* out, int n);
/* performance knobs for A/B measurements; results never depend on them.
 * "euler_march" (default 1): k-marching fused kernel for Euler + scalar JST; the full list with defaults: DESIGN.md section 8b */
int adflow_gpu_set_tuning(const char* key, int value);
/* on != 0: hot-path entry points only ENQUEUE on the library stream (no host
 * sync at return); the caller orders with adflow_gpu_sync().  Default off. */
int adflow_gpu_set_async(int on);
/* sizeof(adflow_opts), sizeof(adflow_block_desc) as compiled: lets a foreign-
 * language binding verify its mirror of the two structs */
int adflow_gpu_abi_sizes(int* opts_bytes, int* desc_bytes);
/* the same for adflow_bc_subface and adflow_comm_pattern */
int adflow_gpu_abi_sizes2(int* bc_subface_bytes, int* comm_pattern_bytes);

/* ---- preconditioner / Jacobian assembly: adjointUtils::setupStateResidualMatrix with useAD = F (src/adjoint/adjointUtils.F90:7-715),
 * consumers NKSolver::FormJacobianNK (NKSolvers.F90:372-435), FormJacobianANK (:1935-2039), the adjoint's dRdwT.
 * Coloured finite differences of the level's residual (block_res_state, masterRoutines.F90:1214-1283: closures incl. halos,
 * boundary conditions, residual core, resScale) on the device: one residual evaluation per colour and state variable fills one
 * column of every stencil block of the matrix.
 *   ADFLOW_JAC_PC          usePC: 7-point stencil, lumped dissipation with the frozen sensor, thin-layer viscous flux,
 *                          first-order turbulence advection, acousticScaleFactor = 1 (7 colours)
 *   without it             the exact dR/dw: 13-point stencil / 13 colours (Euler), 33-point stencil / 35 colours (viscous)
 *   ADFLOW_JAC_FROZEN_TURB frozenTurb: nState = nwf, RANS evaluated as laminar NS plus the eddy viscosity, no SA residual
 *   ADFLOW_JAC_TURB_ONLY   useTurbOnly (the turbulence KSP of ANK, NKSolvers.F90:2340-2370): nState = 1, only the SA residual
 *   ADFLOW_JAC_VISC_PC     inputAdjoint::viscPC with ADFLOW_JAC_PC: the 27-point stencil and the 3x3x3 colouring
 * delta: the finite-difference step (the reference uses 1e-9).  The state is restored afterwards, dw holds the scaled reference
 * residual (resetFDReference).  level must be the ground level.
 *   ADFLOW_JAC_USE_AD      useAD = T (adjointUtils.F90:227-409): every column from ONE forward-mode evaluation (seed 1 on the state
 *                          variable of the colour's cells, masterRoutines::block_res_state_d) instead of a finite difference: the
 *                          exact derivative, `delta` is not used.  Dual-number twins of the gather kernels (csrc/kernels_ad.hip).
 *                          The dual copies of the level's arrays (about 640 B per box cell) are one slab that is KEPT between calls
 *                          and freed with the blocks (adflow_gpu_block_release / _release_all) or by tuning "ad_cache" = 0; the call
 *                          fails with a message when the device has not that much memory free.
 */
enum { ADFLOW_JAC_PC = 1u, ADFLOW_JAC_FROZEN_TURB = 2u, ADFLOW_JAC_TURB_ONLY = 4u, ADFLOW_JAC_VISC_PC = 8u, ADFLOW_JAC_USE_AD = 16u };
int adflow_gpu_fd_jacobian(int level, unsigned flags, double delta);
/* Hands back the work space an assembly keeps between calls -- the slab of dual arrays of ADFLOW_JAC_USE_AD (about 640 B per box cell:
 * 8.4 GB on the 8 x 160x128x64 mesh) -- to the device allocator: what the host calls when the matrix is assembled and the memory is
 * wanted elsewhere (PETSc objects, further multigrid levels).  *bytes (may be NULL): what was released.  The next forward-mode
 * assembly lays the slab out again.  Mirrors nothing in the reference, whose Tapenade derivative arrays live in flowDomsd for the
 * whole run (adjointUtils.F90:87-99 allocDerivativeValues). */
int adflow_gpu_release_workspace(int64_t* bytes);
/* Self-test of the arithmetic the kernels substitute for the compiler's division, square root, pow and exp (csrc/internal.h: v_rcp_f64 /
 * v_rsq_f64 seeds + Newton steps, x^(1/6), x^a, exp of a negative argument) and of their dual-number forms (csrc/kernels_ad.hip): for
 * every i < n   y[i] = f(x[i])  by the plain form and  dy[2i], dy[2i+1] = value and d/dx by the dual form.  which: 0 1/x, 1 1/sqrt(x),
 * 2 sqrt(x), 3 x^(1/6), 4 exp(x) for x <= 0, 5 x^a[i], 6 a[i]/x, 7 x/a[i].  Host pointers.  Mirrors nothing in the reference (its
 * compiler's own division and intrinsics, e.g. sa.F90:245-330, solverUtils.F90:292-310): the check that the substitution stays inside
 * the parity bar on the arguments the flow kernels see, which the CPU emulator of the tests cannot make (it runs libm). */
int adflow_gpu_selftest_math(int which, const double* x, const double* a, int64_t n, double* y, double* dy);
/* nState, nStencil and the stencil offsets (nStencil,3) column-major as src/modules/stencils.f90 of the last assembly:
 * block (ll, l) of stencil entry s at row cell (i,j,k) is  d dw(i,j,k,ll) / d w(i-di(s), j-dj(s), k-dk(s), l)  (after resScale) */
int adflow_gpu_jacobian_info(int32_t* nState, int32_t* nStencil, int32_t* stencil);
/* blocks of block nn over its OWNED cells: (nx, ny, nz, nState, nState, nStencil) column-major.  The host maps rows / columns to
 * globalCell and calls MatSetValuesBlocked; entries whose source cell lies outside 0..ib are zero */
int adflow_gpu_download_jacobian(int nn, int level, int sps, double* blocks);
/* the same blocks in the order of the reference's insertion loop (adjointUtils.F90:560-700, one MatSetValuesBlocked per row cell
 * and stencil entry with MAT_ROW_ORIENTED off): rows(nState, nState, nStencil, nx, ny, nz) column-major, i.e. the nStencil
 * blocks blk(ll, l) of a row cell are contiguous and the cells follow in the order of the PETSc rows of the block (i fastest).
 * Transposed on the device, copied in slabs of k planes */
int adflow_gpu_download_jacobian_rows(int nn, int level, int sps, double* rows);

#ifdef __cplusplus
}
#endif
#endif /* ADFLOW_GPU_H */
