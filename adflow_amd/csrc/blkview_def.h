// The device view of one block.  No include guard: internal.h includes this with ADF_BLKVIEW = BlkView; the forward-AD translation
// unit (kernels_ad.hip) includes it a second time with `double` standing for the dual number type and ADF_BLKVIEW = BlkViewAD --
// the same fields at the same offsets (every pointer has the size of a pointer, the two value fields are adf_real8 = a real double
// in both), so that the host fills a BlkViewAD through the BlkView layout.
struct ADF_BLKVIEW {
    int nx, ny, nz, nw;
    int il, jl, kl, ie, je, ke, ib, jb, kb;
    int ldi;        // j-stride (doubles)
    int ldk;        // k-stride
    long nbox;      // stride between variables of a multi-component array
    // state
    double *w, *p, *gamma, *rlv, *rev;
    // geometry
    // (ADF_GEOM = double; in the forward-mode build the geometry stays PLAIN -- adf_real8 -- and points at the library's own arrays: it
    //  carries no derivative, and a dual copy of it would double its bytes)
    ADF_GEOM *x, *sI, *sJ, *sK, *vol, *volRef, *d2wall;
    ADF_GEOM* sFace;        // moving blocks: sFaceI/J/K as components 0..2 (entry at the left cell of the face); NULL at rest
    int moving;             // blockIsMoving: rotational source with rot = cgnsDoms%rotRate (fluxes.F90:372-397)
    adf_real8 rot[3];
    ADF_GEOM *dI, *dJ, *dK; // derived geometry: vector between the two cell centres of a face (viscous normal correction)
    ADF_GEOM* xc;           // derived geometry: the cell centres (mean of the eight corner nodes), cells 1..ie x 1..je x 1..ke (k_visc_gf)
    // implicit turbulence boundary treatment of Spalart-Allmaras (turbBCRoutines.F90:662-798): halo = bvt - bmt * interior.
    // Index 0..5 = iMin,iMax,jMin,jMax,kMin,kMax; entry (a,b) at (a-1) + A*(b-1), A = je (i faces) or ie (j,k faces).
    // NULL until a block registers boundary subfaces (= all zero, the periodic / internal case).
    double *bmt[6], *bvt[6];
    uint8_t* flags;  // bits 0-1 porI+1, 2-3 porJ+1, 4-5 porK+1, bit 6 iblank>0
    // residual + work
    double *dw, *fw, *dtl, *radI, *radJ, *radK;
    double *ss;      // JST sensor variable (entropy p/rho^gamma) for NS/RANS
    double *aa;      // speed of sound squared
    double *grad;    // 12 nodal gradients ux,uy,uz,vx,...,qz
    double *scratch; // nscratch work arrays
    double *wn, *pn; // RK stage-0 state
    double *w1, *p1, *wr; // multigrid
    // multigrid maps (device copies of coarseUtils.F90:254-262): index 2*m+{0,1} for coarse/fine cell m
    int *mgIFine, *mgJFine, *mgKFine;        // coarse block: (1:ie,2) stored [m*2+q], m = 0..ie
    double *mgIWeight, *mgJWeight, *mgKWeight;  // coarse block, indexed by cell index
    int *mgICoarse, *mgJCoarse, *mgKCoarse;  // fine block, indexed [i*2+q]
    adf_real8 mfact;    // +0.5 (right-handed block) or -0.5: the factor of the face-normal cross products (metric_block, adjointExtra.F90:176-268)
    long vecOff;     // first entry of the block in the PETSc-ordered state / residual vector of its level (NKSolvers.F90:1240-1253)
    __host__ __device__ inline long idx(int i, int j, int k) const { return (long)i + (long)j * ldi + (long)k * ldk; }
};
