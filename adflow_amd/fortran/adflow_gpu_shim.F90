! adflow_gpu_shim.F90 — ISO_C_BINDING interface to the C-ABI of include/adflow_gpu.h
! plus the glue that fills its two structs from the reference's own modules.
!
! This file is what a maintainer ADDS to the reference tree (src/gpu/, listed in
! src/build/fileList after modules/ and utils/); it `use`s the reference's
! modules and therefore only compiles inside that tree (oracle/refbuild proves
! it does: `make -C oracle/refbuild shim` compiles it against the reference's
! .mod files with amdflang).  INTEGRATION.md shows the call-site edits.
module adflowGpuShim
    use iso_c_binding
    use constants
    implicit none

    ! ---- flag bits of adflow_gpu_block_res / adflow_gpu_fd_jacobian (include/adflow_gpu.h) ----
    integer(c_int), parameter :: ADFLOW_RES_UPDATE_INTERMED = 1, ADFLOW_RES_FLOW = 2, ADFLOW_RES_TURB = 4, ADFLOW_RES_CLOSURES = 8, &
                                 ADFLOW_RES_HALO = 16, ADFLOW_RES_DISS_APPROX = 32, ADFLOW_RES_VISC_APPROX = 64, &
                                 ADFLOW_RES_UPWIND_FIRST_ORDER = 128
    integer(c_int), parameter :: ADFLOW_JAC_PC = 1, ADFLOW_JAC_FROZEN_TURB = 2, ADFLOW_JAC_TURB_ONLY = 4, ADFLOW_JAC_VISC_PC = 8, &
                                 ADFLOW_JAC_USE_AD = 16

    ! ---- mirror of adflow_opts (include/adflow_gpu.h) -----------------------
    type, bind(C) :: adflow_opts
        integer(c_int32_t) :: equations, turbModel, turbProd
        integer(c_int32_t) :: useQCR, useRotationSA, useft2SA
        integer(c_int32_t) :: spaceDiscr, spaceDiscrCoarse, limiter, orderTurb
        integer(c_int32_t) :: dirScaling
        integer(c_int32_t) :: smoother, nRKStages, resAveraging, nSubiterations, nSubIterTurb
        integer(c_int32_t) :: groundLevel
        integer(c_int32_t) :: turbRelax
        integer(c_int32_t) :: eulerWallBCTreatment, viscWallBCTreatment, outflowTreatment
        integer(c_int32_t) :: hScalingInlet, unsupported
        integer(c_int32_t) :: lowSpeedPreconditioner
        integer(c_int32_t) :: exchangePressureEarly, reserved_i
        real(c_double) :: gammaConstant, prandtl, prandtlTurb
        real(c_double) :: SSuthDim, muSuthDim, TSuthDim
        real(c_double) :: SAKappa, SAcb1, SAcb2, SAsigma, SAcv1, SAcw1, SAcw2, SAcw3, SAct1, SAct2, SAct3, SAct4, SAcrot
        real(c_double) :: vis2, vis4, vis2Coarse, adis, acousticScaleFactor, kappaCoef
        real(c_double) :: cfl, cflCoarse, cflLimit, fcoll, smoop, alfaTurb, betaTurb, turbResScale
        real(c_double) :: etaRK(8), cdisRK(8)
        real(c_double) :: gammaInf, pInf, pInfCorr, rhoInf, uInf, RGas, muInf, muRef, TRef, timeRef
        real(c_double) :: wInf(10)
        real(c_double) :: sigma
        real(c_double) :: pRef, uRef, LRef, ordersConverged
        real(c_double) :: reserved_d(3)
    end type adflow_opts

    ! ---- mirror of adflow_block_desc ----------------------------------------
    type, bind(C) :: adflow_block_desc
        integer(c_int32_t) :: nx, ny, nz, nw, rightHanded, reserved
        type(c_ptr) :: w, p, gamma, rlv, rev
        type(c_ptr) :: x, sI, sJ, sK, vol, volRef, d2Wall
        type(c_ptr) :: porI, porJ, porK, iblank
        type(c_ptr) :: dw, fw, dtl, radI, radJ, radK
        type(c_ptr) :: w1, p1, wr
        type(c_ptr) :: mgIFine, mgJFine, mgKFine, mgIWeight, mgJWeight, mgKWeight
        type(c_ptr) :: mgICoarse, mgJCoarse, mgKCoarse
        type(c_ptr) :: sFaceI, sFaceJ, sFaceK
        real(c_double) :: rotRate(3)
        integer(c_int32_t) :: addGridVelocities, blockIsMoving
    end type adflow_block_desc

    ! ---- mirror of adflow_comm_pattern ---------------------------------------
    type, bind(C) :: adflow_comm_pattern
        integer(c_int32_t) :: ncopy
        type(c_ptr) :: donorBlock, donorIndices, haloBlock, haloIndices
        integer(c_int32_t) :: nProcSend
        type(c_ptr) :: sendProc, nsendCum, sendBlock, sendIndices
        integer(c_int32_t) :: nProcRecv
        type(c_ptr) :: recvProc, nrecvCum, recvBlock, recvIndices
    end type adflow_comm_pattern

    ! ---- mirror of adflow_actuator_region --------------------------------------
    type, bind(C) :: adflow_actuator_region
        integer(c_int32_t) :: nCellIDs, reserved
        type(c_ptr) :: block, cellIDs
        real(c_double) :: force(3), heat, volume, relaxStart, relaxEnd
    end type adflow_actuator_region

    type intBuf
        integer(c_int32_t), allocatable :: v(:)
    end type intBuf

    ! ---- mirror of adflow_periodic_data ----------------------------------------
    type, bind(C) :: adflow_periodic_data
        real(c_double) :: rotMatrix(9), rotCenter(3), translation(3)
        integer(c_int32_t) :: nHalos, reserved
        type(c_ptr) :: block, indices
    end type adflow_periodic_data

    ! ---- mirror of adflow_bc_subface ------------------------------------------
    type, bind(C) :: adflow_bc_subface
        integer(c_int32_t) :: bcType, faceID
        integer(c_int32_t) :: icBeg, icEnd, jcBeg, jcEnd
        integer(c_int32_t) :: subsonicInletTreatment, reserved
        type(c_ptr) :: norm, rface, uSlip, TNS_Wall
        type(c_ptr) :: rho, velx, vely, velz, ps
        type(c_ptr) :: ptInlet, ttInlet, htInlet, flowXdirInlet, flowYdirInlet, flowZdirInlet, turbInlet
        real(c_double) :: symNorm(3)
    end type adflow_bc_subface

    interface
        integer(c_int) function adflow_gpu_bc_register(nn, level, sps, nBocos, nViscBocos, faces) &
            bind(C, name="adflow_gpu_bc_register")
            import :: c_int, adflow_bc_subface
            integer(c_int), value :: nn, level, sps, nBocos, nViscBocos
            type(adflow_bc_subface), intent(in) :: faces(*)
        end function
        integer(c_int) function adflow_gpu_upload_coordinates(nn, level, sps) bind(C, name="adflow_gpu_upload_coordinates")
            import :: c_int
            integer(c_int), value :: nn, level, sps
        end function
        integer(c_int) function adflow_gpu_update_geometry(level) bind(C, name="adflow_gpu_update_geometry")
            import :: c_int
            integer(c_int), value :: level
        end function
        integer(c_int) function adflow_gpu_apply_all_bc(level, secondHalo) bind(C, name="adflow_gpu_apply_all_bc")
            import :: c_int
            integer(c_int), value :: level, secondHalo
        end function
        integer(c_int) function adflow_gpu_comm_register_periodic(level, nLayers, nPeriodic, pd) &
            bind(C, name="adflow_gpu_comm_register_periodic")
            import :: c_int, adflow_periodic_data
            integer(c_int), value :: level, nLayers, nPeriodic
            type(adflow_periodic_data), intent(in) :: pd(*)
        end function
        integer(c_int) function adflow_gpu_actuator_register(nRegions, regions) bind(C, name="adflow_gpu_actuator_register")
            import :: c_int, adflow_actuator_region
            integer(c_int), value :: nRegions
            type(adflow_actuator_region), intent(in) :: regions(*)
        end function
        integer(c_int) function adflow_gpu_xhalo(level) bind(C, name="adflow_gpu_xhalo")
            import :: c_int
            integer(c_int), value :: level
        end function
        integer(c_int) function adflow_gpu_coarse_coordinates(coarseLevel) bind(C, name="adflow_gpu_coarse_coordinates")
            import :: c_int
            integer(c_int), value :: coarseLevel
        end function
        integer(c_int) function adflow_gpu_exchange_coor(level) bind(C, name="adflow_gpu_exchange_coor")
            import :: c_int
            integer(c_int), value :: level
        end function
        integer(c_int) function adflow_gpu_download_wall_stress(nn, level, sps, mm, tau, q) &
            bind(C, name="adflow_gpu_download_wall_stress")
            import :: c_int, c_ptr
            integer(c_int), value :: nn, level, sps, mm
            type(c_ptr), value :: tau, q
        end function
        integer(c_int) function adflow_gpu_comm_register(level, nLayers, p) bind(C, name="adflow_gpu_comm_register")
            import :: c_int, adflow_comm_pattern
            integer(c_int), value :: level, nLayers
            type(adflow_comm_pattern), intent(in) :: p
        end function
        integer(c_int) function adflow_gpu_init(device) bind(C, name="adflow_gpu_init")
            import :: c_int
            integer(c_int), value :: device
        end function
        integer(c_int) function adflow_gpu_finalize() bind(C, name="adflow_gpu_finalize")
            import :: c_int
        end function
        type(c_ptr) function adflow_gpu_last_error() bind(C, name="adflow_gpu_last_error")
            import :: c_ptr
        end function
        integer(c_int) function adflow_gpu_set_options(o) bind(C, name="adflow_gpu_set_options")
            import :: c_int, adflow_opts
            type(adflow_opts), intent(in) :: o
        end function
        integer(c_int) function adflow_gpu_block_register(nn, level, sps, d) bind(C, name="adflow_gpu_block_register")
            import :: c_int, adflow_block_desc
            integer(c_int), value :: nn, level, sps
            type(adflow_block_desc), intent(in) :: d
        end function
        integer(c_int) function adflow_gpu_upload_geometry(nn, level, sps) bind(C, name="adflow_gpu_upload_geometry")
            import :: c_int
            integer(c_int), value :: nn, level, sps
        end function
        integer(c_int) function adflow_gpu_upload_state(nn, level, sps) bind(C, name="adflow_gpu_upload_state")
            import :: c_int
            integer(c_int), value :: nn, level, sps
        end function
        integer(c_int) function adflow_gpu_download_state(nn, level, sps) bind(C, name="adflow_gpu_download_state")
            import :: c_int
            integer(c_int), value :: nn, level, sps
        end function
        integer(c_int) function adflow_gpu_download_residual(nn, level, sps) bind(C, name="adflow_gpu_download_residual")
            import :: c_int
            integer(c_int), value :: nn, level, sps
        end function
        integer(c_int) function adflow_gpu_time_step(level, onlyRadii) bind(C, name="adflow_gpu_time_step")
            import :: c_int
            integer(c_int), value :: level, onlyRadii
        end function
        integer(c_int) function adflow_gpu_initres(level, varStart, varEnd) bind(C, name="adflow_gpu_initres")
            import :: c_int
            integer(c_int), value :: level, varStart, varEnd
        end function
        integer(c_int) function adflow_gpu_residual(level, rkStage) bind(C, name="adflow_gpu_residual")
            import :: c_int
            integer(c_int), value :: level, rkStage
        end function
        integer(c_int) function adflow_gpu_block_res(level, flags) bind(C, name="adflow_gpu_block_res")
            import :: c_int
            integer(c_int), value :: level, flags
        end function
        integer(c_int) function adflow_gpu_rk_smooth(level) bind(C, name="adflow_gpu_rk_smooth")
            import :: c_int
            integer(c_int), value :: level
        end function
        integer(c_int) function adflow_gpu_dadi_smooth(level) bind(C, name="adflow_gpu_dadi_smooth")
            import :: c_int
            integer(c_int), value :: level
        end function
        integer(c_int) function adflow_gpu_halo_exchange(level, varStart, varEnd, commPressure, commVisc, nLayers) &
            bind(C, name="adflow_gpu_halo_exchange")
            import :: c_int
            integer(c_int), value :: level, varStart, varEnd, commPressure, commVisc, nLayers
        end function
        integer(c_int) function adflow_gpu_res_norms(level, sums, n) bind(C, name="adflow_gpu_res_norms")
            import :: c_int, c_double
            integer(c_int), value :: level, n
            real(c_double), intent(out) :: sums(*)
        end function
        integer(c_int) function adflow_gpu_mg_cycle(cycling, nStepsCycling) bind(C, name="adflow_gpu_mg_cycle")
            import :: c_int, c_int32_t
            integer(c_int32_t), intent(in) :: cycling(*)
            integer(c_int), value :: nStepsCycling
        end function
        integer(c_int) function adflow_gpu_transfer_to_coarse(level) bind(C, name="adflow_gpu_transfer_to_coarse")
            import :: c_int
            integer(c_int), value :: level
        end function
        integer(c_int) function adflow_gpu_transfer_to_fine(level) bind(C, name="adflow_gpu_transfer_to_fine")
            import :: c_int
            integer(c_int), value :: level
        end function
        integer(c_int) function adflow_gpu_sa_solve(level) bind(C, name="adflow_gpu_sa_solve")
            import :: c_int
            integer(c_int), value :: level
        end function
        integer(c_int) function adflow_gpu_wall_distance_register(nn, level, sps, surfNodeIndices, uv) &
            bind(C, name="adflow_gpu_wall_distance_register")
            import :: c_int, c_ptr
            integer(c_int), value :: nn, level, sps
            type(c_ptr), value :: surfNodeIndices, uv
        end function
        integer(c_int) function adflow_gpu_update_wall_distances(level, xSurf, n) bind(C, name="adflow_gpu_update_wall_distances")
            import :: c_int, c_ptr, c_int64_t
            integer(c_int), value :: level
            type(c_ptr), value :: xSurf
            integer(c_int64_t), value :: n
        end function
        ! RCCL bootstrap (gpuCommInit below), host hooks, device-resident NK vectors, tuning
        integer(c_int) function adflow_gpu_comm_unique_id(id128) bind(C, name="adflow_gpu_comm_unique_id")
            import :: c_int, c_char
            character(kind=c_char), intent(out) :: id128(128)
        end function
        integer(c_int) function adflow_gpu_comm_init(rank, nranks, id128) bind(C, name="adflow_gpu_comm_init")
            import :: c_int, c_char
            integer(c_int), value :: rank, nranks
            character(kind=c_char), intent(in) :: id128(128)
        end function
        integer(c_int) function adflow_gpu_comm_info(rank, nranks, commCount, commUserRank) bind(C, name="adflow_gpu_comm_info")
            import :: c_int
            integer(c_int), intent(out) :: rank, nranks, commCount, commUserRank
        end function
        integer(c_int) function adflow_gpu_set_bc_callback(fn) bind(C, name="adflow_gpu_set_bc_callback")
            import :: c_int, c_funptr
            type(c_funptr), value :: fn        ! subroutine fn(level, secondHalo) bind(C), integer(c_int), value arguments
        end function
        integer(c_int) function adflow_gpu_set_turb_bc_callback(fn) bind(C, name="adflow_gpu_set_turb_bc_callback")
            import :: c_int, c_funptr
            type(c_funptr), value :: fn
        end function
        integer(c_int) function adflow_gpu_nk_residual_dev(wVec, rVec, n) bind(C, name="adflow_gpu_nk_residual_dev")
            import :: c_int, c_ptr, c_long
            type(c_ptr), value :: wVec, rVec   ! DEVICE pointers (VecHIPGetArray of PETSc vectors living on the GPU)
            integer(c_long), value :: n
        end function
        integer(c_int) function adflow_gpu_set_tuning(key, value) bind(C, name="adflow_gpu_set_tuning")
            import :: c_int, c_char
            character(kind=c_char), intent(in) :: key(*)
            integer(c_int), value :: value
        end function
        ! adjointUtils::setupStateResidualMatrix: coloured finite-difference blocks (useAD = F) or, with ADFLOW_JAC_USE_AD, the
        ! forward-mode blocks of useAD = T (adjointUtils.F90:227-409), on the device
        integer(c_int) function adflow_gpu_fd_jacobian(level, flags, delta) bind(C, name="adflow_gpu_fd_jacobian")
            import :: c_int, c_double
            integer(c_int), value :: level, flags
            real(c_double), value :: delta
        end function
        ! the work space a forward-mode assembly keeps between calls (the slab of dual arrays) back to the device allocator
        integer(c_int) function adflow_gpu_release_workspace(bytes) bind(C, name="adflow_gpu_release_workspace")
            import :: c_int, c_ptr
            type(c_ptr), value :: bytes
        end function
        ! self-test of the fast division / root / power forms of the kernels (plain and dual-number)
        integer(c_int) function adflow_gpu_selftest_math(which, x, a, n, y, dy) bind(C, name="adflow_gpu_selftest_math")
            import :: c_int, c_int64_t, c_ptr
            integer(c_int), value :: which
            type(c_ptr), value :: x, a, y, dy
            integer(c_int64_t), value :: n
        end function
        integer(c_int) function adflow_gpu_jacobian_info(nState, nStencil, stencil) bind(C, name="adflow_gpu_jacobian_info")
            import :: c_int, c_ptr
            integer(c_int), intent(out) :: nState, nStencil
            type(c_ptr), value :: stencil
        end function
        integer(c_int) function adflow_gpu_download_jacobian(nn, level, sps, blocks) bind(C, name="adflow_gpu_download_jacobian")
            import :: c_int, c_ptr
            integer(c_int), value :: nn, level, sps
            type(c_ptr), value :: blocks
        end function
        integer(c_int) function adflow_gpu_download_jacobian_rows(nn, level, sps, rows) bind(C, name="adflow_gpu_download_jacobian_rows")
            import :: c_int, c_ptr
            integer(c_int), value :: nn, level, sps
            type(c_ptr), value :: rows
        end function
        integer(c_int) function adflow_gpu_reference_shock_sensor(level) bind(C, name="adflow_gpu_reference_shock_sensor")
            import :: c_int
            integer(c_int), value :: level
        end function
        integer(c_int) function adflow_gpu_set_w_vec(wVec, n) bind(C, name="adflow_gpu_set_w_vec")
            import :: c_int, c_long, c_double
            real(c_double), intent(in) :: wVec(*)
            integer(c_long), value :: n
        end function
        integer(c_int) function adflow_gpu_get_r_vec(rVec, n, sumsq2) bind(C, name="adflow_gpu_get_r_vec")
            import :: c_int, c_long, c_double, c_ptr
            real(c_double), intent(out) :: rVec(*)
            integer(c_long), value :: n
            type(c_ptr), value :: sumsq2
        end function
        integer(c_int) function adflow_gpu_get_res(res, n) bind(C, name="adflow_gpu_get_res")
            import :: c_int, c_long, c_double
            real(c_double), intent(out) :: res(*)
            integer(c_long), value :: n
        end function
        integer(c_int) function adflow_gpu_nk_residual(wVec, rVec, n) bind(C, name="adflow_gpu_nk_residual")
            import :: c_int, c_long, c_double
            real(c_double), intent(in) :: wVec(*)
            real(c_double), intent(out) :: rVec(*)
            integer(c_long), value :: n
        end function
        integer(c_int) function adflow_gpu_block_release(nn, level, sps) bind(C, name="adflow_gpu_block_release")
            import :: c_int
            integer(c_int), value :: nn, level, sps
        end function
        integer(c_int) function adflow_gpu_release_all() bind(C, name="adflow_gpu_release_all")
            import :: c_int
        end function
        integer(c_int) function adflow_gpu_sync() bind(C, name="adflow_gpu_sync")
            import :: c_int
        end function
        integer(c_int) function adflow_gpu_set_async(on) bind(C, name="adflow_gpu_set_async")
            import :: c_int
            integer(c_int), value :: on
        end function
    end interface

contains

    ! forward a library error to the reference's own error path (utils.F90:501)
    subroutine gpuCheck(ierr, routine)
        use utils, only: terminate
        integer(c_int), intent(in) :: ierr
        character(len=*), intent(in) :: routine
        character(kind=c_char), pointer :: cmsg(:)
        character(len=512) :: msg
        integer :: i
        if (ierr == 0) return
        call c_f_pointer(adflow_gpu_last_error(), cmsg, [512])
        msg = ' '
        do i = 1, 512
            if (cmsg(i) == c_null_char) exit
            msg(i:i) = cmsg(i)
        end do
        call terminate(routine, trim(msg))
    end subroutine gpuCheck

    ! snapshot of the module variables the hot path reads (what Python may have
    ! reassigned through f2py since the last call, pyADflow.py:5463-5630)
    subroutine gpuRefreshOptions()
        use inputPhysics
        use inputDiscretization
        use inputIteration
        use iteration, only: groundLevel, ordersConverged, exchangePressureEarly
        use flowVarRefState
        use paramTurb, only: rsaCw1
        use oversetData, only: oversetPresent
        type(adflow_opts) :: o
        integer :: n
        o%equations = equations; o%turbModel = turbModel; o%turbProd = turbProd
        o%useQCR = merge(1, 0, useQCR); o%useRotationSA = merge(1, 0, useRotationSA); o%useft2SA = merge(1, 0, useft2SA)
        o%spaceDiscr = spaceDiscr; o%spaceDiscrCoarse = spaceDiscrCoarse; o%limiter = limiter; o%orderTurb = orderTurb
        o%dirScaling = merge(1, 0, dirScaling)
        o%smoother = smoother; o%nRKStages = nRKStages; o%resAveraging = resAveraging
        o%nSubiterations = nSubiterations; o%nSubIterTurb = nSubIterTurb
        o%groundLevel = groundLevel
        o%turbRelax = turbRelax
        o%eulerWallBCTreatment = eulerWallBCTreatment; o%viscWallBCTreatment = viscWallBCTreatment
        o%outflowTreatment = outflowTreatment
        o%hScalingInlet = merge(1, 0, hScalingInlet)
        ! configurations outside the path: refused by the library (adflow_gpu_set_options) instead of computed wrongly
        o%unsupported = 0
        if (equationMode /= steady) o%unsupported = ior(o%unsupported, 1)
        if (cpModel /= cpConstant) o%unsupported = ior(o%unsupported, 2)
        if (wallFunctions) o%unsupported = ior(o%unsupported, 4)
        if (oversetPresent) o%unsupported = ior(o%unsupported, 8)
        o%lowSpeedPreconditioner = merge(1, 0, lowSpeedPreconditioner)
        o%exchangePressureEarly = merge(1, 0, exchangePressureEarly)
        o%reserved_i = 0
        o%gammaConstant = gammaConstant; o%prandtl = prandtl; o%prandtlTurb = prandtlTurb
        o%SSuthDim = SSuthDim; o%muSuthDim = muSuthDim; o%TSuthDim = TSuthDim
        o%SAKappa = SAKappa; o%SAcb1 = SAcb1; o%SAcb2 = SAcb2; o%SAsigma = SAsigma; o%SAcv1 = SAcv1
        o%SAcw1 = rsaCw1; o%SAcw2 = SAcw2; o%SAcw3 = SAcw3
        o%SAct1 = SAct1; o%SAct2 = SAct2; o%SAct3 = SAct3; o%SAct4 = SAct4; o%SAcrot = SAcrot
        o%vis2 = vis2; o%vis4 = vis4; o%vis2Coarse = vis2Coarse; o%adis = adis
        o%acousticScaleFactor = acousticScaleFactor; o%kappaCoef = kappaCoef
        o%cfl = cfl; o%cflCoarse = cflCoarse; o%cflLimit = cflLimit; o%fcoll = fcoll; o%smoop = smoop
        o%alfaTurb = alfaTurb; o%betaTurb = betaTurb; o%turbResScale = turbResScale(1)
        o%etaRK = zero; o%cdisRK = zero
        n = min(nRKStages, 8)
        if (allocated(etaRK)) o%etaRK(1:n) = etaRK(1:n)
        if (allocated(cdisRK)) o%cdisRK(1:n) = cdisRK(1:n)
        o%gammaInf = gammaInf; o%pInf = pInf; o%pInfCorr = pInfCorr; o%rhoInf = rhoInf; o%uInf = uInf
        o%RGas = RGas; o%muInf = muInf; o%muRef = muRef; o%TRef = TRef; o%timeRef = timeRef
        o%wInf = zero
        if (allocated(wInf)) o%wInf(1:size(wInf)) = wInf
        o%sigma = sigma
        o%pRef = pRef; o%uRef = uRef; o%LRef = LRef; o%ordersConverged = ordersConverged
        o%reserved_d = zero
        call gpuCheck(adflow_gpu_set_options(o), "gpuRefreshOptions")
    end subroutine gpuRefreshOptions

    ! flowDoms(nn,level,sps) -> device mirror.  All targets are contiguous
    ! `allocate`d arrays (SURVEY.md §8(a) row T) so c_loc is legal.
    subroutine gpuRegisterBlock(nn, level, sps)
        use block, only: flowDoms
        use cgnsGrid, only: cgnsDoms
        use flowVarRefState, only: nw
        integer(kind=intType), intent(in) :: nn, level, sps
        type(adflow_block_desc) :: d
        associate (b => flowDoms(nn, level, sps), b1 => flowDoms(nn, 1, sps), g => flowDoms(nn, level, 1))
            d%nx = b%nx; d%ny = b%ny; d%nz = b%nz; d%nw = nw
            d%rightHanded = merge(1, 0, b%rightHanded); d%reserved = 0
            d%w = c_loc(b%w); d%p = c_loc(b%p)
            ! gamma, rlv, dw, fw, dtl, radI/J/K exist on the finest level only; setPointers aims the coarse levels at the
            ! FINE arrays and indexes them with coarse i,j,k (utils.F90:3310-3360), i.e. with the fine strides.  The library
            ! copies contiguous level-shaped boxes, so the coarse levels do not hand these over (gamma is the constant of
            ! the calorically perfect gas there, the others are device-only work arrays on coarse levels).
            d%gamma = c_null_ptr; d%rlv = c_null_ptr; d%rev = c_null_ptr
            if (level == 1) then
                d%gamma = c_loc(b1%gamma); d%rlv = c_loc(b1%rlv)
            end if
            if (associated(b%rev)) d%rev = c_loc(b%rev)
            d%x = c_loc(b%x); d%sI = c_loc(b%sI); d%sJ = c_loc(b%sJ); d%sK = c_loc(b%sK)
            d%vol = c_loc(b%vol); d%volRef = c_null_ptr; d%d2Wall = c_null_ptr
            if (associated(b%volRef)) d%volRef = c_loc(b%volRef)
            if (associated(b%d2Wall)) d%d2Wall = c_loc(b%d2Wall)
            d%porI = c_loc(g%porI); d%porJ = c_loc(g%porJ); d%porK = c_loc(g%porK)
            d%iblank = c_loc(b%iblank)
            d%dw = c_null_ptr; d%fw = c_null_ptr; d%dtl = c_null_ptr
            d%radI = c_null_ptr; d%radJ = c_null_ptr; d%radK = c_null_ptr
            if (level == 1) then
                d%dw = c_loc(b1%dw); d%fw = c_loc(b1%fw); d%dtl = c_loc(b1%dtl)
                d%radI = c_loc(b1%radI); d%radJ = c_loc(b1%radJ); d%radK = c_loc(b1%radK)
            end if
            d%w1 = c_null_ptr; d%p1 = c_null_ptr; d%wr = c_null_ptr
            d%mgIFine = c_null_ptr; d%mgJFine = c_null_ptr; d%mgKFine = c_null_ptr
            d%mgIWeight = c_null_ptr; d%mgJWeight = c_null_ptr; d%mgKWeight = c_null_ptr
            d%mgICoarse = c_null_ptr; d%mgJCoarse = c_null_ptr; d%mgKCoarse = c_null_ptr
            if (level > 1) then
                d%w1 = c_loc(b%w1); d%p1 = c_loc(b%p1); d%wr = c_loc(b%wr)
                d%mgIFine = c_loc(g%mgIFine); d%mgJFine = c_loc(g%mgJFine); d%mgKFine = c_loc(g%mgKFine)
                d%mgIWeight = c_loc(g%mgIWeight); d%mgJWeight = c_loc(g%mgJWeight); d%mgKWeight = c_loc(g%mgKWeight)
            end if
            if (associated(g%mgICoarse)) then
                d%mgICoarse = c_loc(g%mgICoarse); d%mgJCoarse = c_loc(g%mgJCoarse); d%mgKCoarse = c_loc(g%mgKCoarse)
            end if
            d%sFaceI = c_null_ptr; d%sFaceJ = c_null_ptr; d%sFaceK = c_null_ptr; d%rotRate = 0.0_c_double
            d%addGridVelocities = merge(1, 0, b%addGridVelocities); d%blockIsMoving = merge(1, 0, b%blockIsMoving)
            if (b%addGridVelocities) then
                d%sFaceI = c_loc(b%sFaceI); d%sFaceJ = c_loc(b%sFaceJ); d%sFaceK = c_loc(b%sFaceK)
            end if
            if (b%blockIsMoving) d%rotRate = cgnsDoms(b%cgnsBlockID)%rotRate
        end associate
        call gpuCheck(adflow_gpu_block_register(int(nn, c_int), int(level, c_int), int(sps, c_int), d), "gpuRegisterBlock")
        call gpuCheck(adflow_gpu_upload_geometry(int(nn, c_int), int(level, c_int), int(sps, c_int)), "gpuRegisterBlock")
    end subroutine gpuRegisterBlock

    ! commPatternCell_{1st,2nd}(level) + internalCell_{1st,2nd}(level) -> adflow_comm_pattern
    ! (src/modules/communication.F90).  Pure flattening: index values are passed as stored.
    subroutine gpuRegisterComm(level, nLayers, cp, ic)
        use communication, only: commType, internalCommType
        integer(kind=intType), intent(in) :: level, nLayers
        type(commType), intent(in) :: cp
        type(internalCommType), intent(in) :: ic
        type(adflow_comm_pattern) :: p
        integer(c_int32_t), allocatable, target :: sendBlock(:), sendIdx(:, :), recvBlock(:), recvIdx(:, :)
        integer(c_int32_t), allocatable, target :: nsc(:), nrc(:), sp(:), rp(:)
        integer(c_int32_t), allocatable, target :: dB(:), dI(:, :), hB(:), hI(:, :)
        integer :: i, n0, n1, nst, nrt
        nst = 0; nrt = 0
        if (cp%nProcSend > 0) nst = cp%nsendCum(cp%nProcSend)
        if (cp%nProcRecv > 0) nrt = cp%nrecvCum(cp%nProcRecv)
        allocate (sendBlock(nst), sendIdx(nst, 3), recvBlock(nrt), recvIdx(nrt, 3))
        allocate (nsc(0:cp%nProcSend), nrc(0:cp%nProcRecv), sp(cp%nProcSend), rp(cp%nProcRecv))
        nsc(0) = 0; nrc(0) = 0
        do i = 1, cp%nProcSend
            n0 = cp%nsendCum(i - 1); n1 = cp%nsendCum(i)
            sp(i) = cp%sendProc(i); nsc(i) = n1
            sendBlock(n0 + 1:n1) = cp%sendList(i)%block(1:n1 - n0)
            sendIdx(n0 + 1:n1, :) = cp%sendList(i)%indices(1:n1 - n0, :)
        end do
        do i = 1, cp%nProcRecv
            n0 = cp%nrecvCum(i - 1); n1 = cp%nrecvCum(i)
            rp(i) = cp%recvProc(i); nrc(i) = n1
            recvBlock(n0 + 1:n1) = cp%recvList(i)%block(1:n1 - n0)
            recvIdx(n0 + 1:n1, :) = cp%recvList(i)%indices(1:n1 - n0, :)
        end do
        allocate (dB(ic%ncopy), dI(ic%ncopy, 3), hB(ic%ncopy), hI(ic%ncopy, 3))
        if (ic%ncopy > 0) then
            dB = ic%donorBlock(1:ic%ncopy); dI = ic%donorIndices(1:ic%ncopy, :)
            hB = ic%haloBlock(1:ic%ncopy); hI = ic%haloIndices(1:ic%ncopy, :)
        end if
        p%ncopy = ic%ncopy
        p%donorBlock = c_loc(dB); p%donorIndices = c_loc(dI); p%haloBlock = c_loc(hB); p%haloIndices = c_loc(hI)
        p%nProcSend = cp%nProcSend; p%sendProc = c_loc(sp); p%nsendCum = c_loc(nsc)
        p%sendBlock = c_loc(sendBlock); p%sendIndices = c_loc(sendIdx)
        p%nProcRecv = cp%nProcRecv; p%recvProc = c_loc(rp); p%nrecvCum = c_loc(nrc)
        p%recvBlock = c_loc(recvBlock); p%recvIndices = c_loc(recvIdx)
        call gpuCheck(adflow_gpu_comm_register(int(level, c_int), int(nLayers, c_int), p), "gpuRegisterComm")
        call registerPeriodic()
    contains
        ! periodicData of the internal and of the inter-processor pattern, concatenated (disjoint halos)
        subroutine registerPeriodic()
            type(adflow_periodic_data), allocatable :: pd(:)
            type(intBuf), allocatable, target :: bb(:), ii(:)
            integer :: np, m, k, n
            np = ic%nPeriodic + cp%nPeriodic
            allocate (pd(max(np, 1)), bb(max(np, 1)), ii(max(np, 1)))
            do m = 1, np
                if (m <= ic%nPeriodic) then
                    associate (q => ic%periodicData(m))
                        n = q%nHalos
                        pd(m)%rotMatrix = reshape(q%rotMatrix, [9]); pd(m)%rotCenter = q%rotCenter
                        pd(m)%translation = q%translation
                        allocate (bb(m)%v(max(n, 1)), ii(m)%v(3 * max(n, 1)))
                        do k = 1, n
                            bb(m)%v(k) = int(q%block(k), c_int32_t)
                            ii(m)%v(k) = int(q%indices(k, 1), c_int32_t)
                            ii(m)%v(n + k) = int(q%indices(k, 2), c_int32_t)
                            ii(m)%v(2 * n + k) = int(q%indices(k, 3), c_int32_t)
                        end do
                    end associate
                else
                    associate (q => cp%periodicData(m - ic%nPeriodic))
                        n = q%nHalos
                        pd(m)%rotMatrix = reshape(q%rotMatrix, [9]); pd(m)%rotCenter = q%rotCenter
                        pd(m)%translation = q%translation
                        allocate (bb(m)%v(max(n, 1)), ii(m)%v(3 * max(n, 1)))
                        do k = 1, n
                            bb(m)%v(k) = int(q%block(k), c_int32_t)
                            ii(m)%v(k) = int(q%indices(k, 1), c_int32_t)
                            ii(m)%v(n + k) = int(q%indices(k, 2), c_int32_t)
                            ii(m)%v(2 * n + k) = int(q%indices(k, 3), c_int32_t)
                        end do
                    end associate
                end if
                pd(m)%nHalos = int(n, c_int32_t); pd(m)%reserved = 0
                pd(m)%block = c_loc(bb(m)%v); pd(m)%indices = c_loc(ii(m)%v)
            end do
            call gpuCheck(adflow_gpu_comm_register_periodic(int(level, c_int), int(nLayers, c_int), int(np, c_int), pd), &
                          "gpuRegisterComm (periodic)")
        end subroutine registerPeriodic
    end subroutine gpuRegisterComm

    ! flowDoms(nn,level,sps)%BCType / BCFaceID / BCData(:) -> device (BCRoutines on the device).  Call again after
    ! anything that changes BCData (setBCDataFineGrid / boundary normals after a mesh warp): the data are copied.
    subroutine gpuRegisterBocos(nn, level, sps)
        use block, only: flowDoms
        integer(kind=intType), intent(in) :: nn, level, sps
        type(adflow_bc_subface), allocatable :: f(:)
        integer :: mm, nb
        nb = flowDoms(nn, level, sps)%nBocos
        allocate (f(max(nb, 1)))
        do mm = 1, nb
            associate (d => flowDoms(nn, level, sps)%BCData(mm))
                f(mm)%bcType = flowDoms(nn, level, sps)%BCType(mm)
                f(mm)%faceID = flowDoms(nn, level, sps)%BCFaceID(mm)
                f(mm)%icBeg = d%icBeg; f(mm)%icEnd = d%icEnd; f(mm)%jcBeg = d%jcBeg; f(mm)%jcEnd = d%jcEnd
                f(mm)%norm = c_null_ptr; f(mm)%rface = c_null_ptr; f(mm)%uSlip = c_null_ptr; f(mm)%TNS_Wall = c_null_ptr
                f(mm)%rho = c_null_ptr; f(mm)%velx = c_null_ptr; f(mm)%vely = c_null_ptr; f(mm)%velz = c_null_ptr
                f(mm)%ps = c_null_ptr
                f(mm)%ptInlet = c_null_ptr; f(mm)%ttInlet = c_null_ptr; f(mm)%htInlet = c_null_ptr
                f(mm)%flowXdirInlet = c_null_ptr; f(mm)%flowYdirInlet = c_null_ptr; f(mm)%flowZdirInlet = c_null_ptr
                f(mm)%turbInlet = c_null_ptr
                f(mm)%subsonicInletTreatment = int(d%subsonicInletTreatment, c_int32_t); f(mm)%reserved = 0
                f(mm)%symNorm = d%symNorm
                if (associated(d%norm)) f(mm)%norm = c_loc(d%norm)
                if (associated(d%rface)) f(mm)%rface = c_loc(d%rface)
                if (associated(d%uSlip)) f(mm)%uSlip = c_loc(d%uSlip)
                if (associated(d%TNS_Wall)) f(mm)%TNS_Wall = c_loc(d%TNS_Wall)
                if (associated(d%rho)) f(mm)%rho = c_loc(d%rho)
                if (associated(d%velx)) f(mm)%velx = c_loc(d%velx)
                if (associated(d%vely)) f(mm)%vely = c_loc(d%vely)
                if (associated(d%velz)) f(mm)%velz = c_loc(d%velz)
                if (associated(d%ps)) f(mm)%ps = c_loc(d%ps)
                if (associated(d%ptInlet)) f(mm)%ptInlet = c_loc(d%ptInlet)
                if (associated(d%ttInlet)) f(mm)%ttInlet = c_loc(d%ttInlet)
                if (associated(d%htInlet)) f(mm)%htInlet = c_loc(d%htInlet)
                if (associated(d%flowXdirInlet)) f(mm)%flowXdirInlet = c_loc(d%flowXdirInlet)
                if (associated(d%flowYdirInlet)) f(mm)%flowYdirInlet = c_loc(d%flowYdirInlet)
                if (associated(d%flowZdirInlet)) f(mm)%flowZdirInlet = c_loc(d%flowZdirInlet)
                if (associated(d%turbInlet)) f(mm)%turbInlet = c_loc(d%turbInlet)
            end associate
        end do
        call gpuCheck(adflow_gpu_bc_register(int(nn, c_int), int(level, c_int), int(sps, c_int), int(nb, c_int), &
                                             int(flowDoms(nn, level, sps)%nViscBocos, c_int), f), "gpuRegisterBocos")
    end subroutine gpuRegisterBocos

    ! actuatorRegions(1:nActuatorRegions) -> device (after addActuatorRegion; again when force / heat change)
    subroutine gpuRegisterActuatorRegions()
        use actuatorRegionData, only: actuatorRegions, nActuatorRegions
        use block, only: nDom
        type(adflow_actuator_region), allocatable :: r(:)
        type(intBuf), allocatable, target :: bb(:), cc(:)
        integer :: m, nn, ii, n
        allocate (r(max(nActuatorRegions, 1)), bb(max(nActuatorRegions, 1)), cc(max(nActuatorRegions, 1)))
        do m = 1, nActuatorRegions
            n = actuatorRegions(m)%nCellIDs
            allocate (bb(m)%v(max(n, 1)), cc(m)%v(3 * max(n, 1)))
            do nn = 1, nDom
                do ii = actuatorRegions(m)%blkPtr(nn - 1) + 1, actuatorRegions(m)%blkPtr(nn)
                    bb(m)%v(ii) = int(nn, c_int32_t)
                end do
            end do
            do ii = 1, n
                cc(m)%v(3 * ii - 2:3 * ii) = int(actuatorRegions(m)%cellIDs(1:3, ii), c_int32_t)
            end do
            r(m)%nCellIDs = int(n, c_int32_t); r(m)%reserved = 0
            r(m)%block = c_loc(bb(m)%v); r(m)%cellIDs = c_loc(cc(m)%v)
            r(m)%force = actuatorRegions(m)%force; r(m)%heat = actuatorRegions(m)%heat
            r(m)%volume = actuatorRegions(m)%volume
            r(m)%relaxStart = actuatorRegions(m)%relaxStart; r(m)%relaxEnd = actuatorRegions(m)%relaxEnd
        end do
        call gpuCheck(adflow_gpu_actuator_register(int(nActuatorRegions, c_int), r), "gpuRegisterActuatorRegions")
    end subroutine gpuRegisterActuatorRegions

    ! viscSubface(:)%tau / %q of a block <- device (what viscousFlux stored with storeWallTensor); call before the host's
    ! force integration (surfaceIntegrations.F90:718) or computeUtau
    subroutine gpuDownloadWallStress(nn, level, sps)
        use block, only: flowDoms
        integer(kind=intType), intent(in) :: nn, level, sps
        integer :: mm
        do mm = 1, flowDoms(nn, level, sps)%nViscBocos
            call gpuCheck(adflow_gpu_download_wall_stress(int(nn, c_int), int(level, c_int), int(sps, c_int), int(mm, c_int), &
                                                          c_loc(flowDoms(nn, level, sps)%viscSubface(mm)%tau), &
                                                          c_loc(flowDoms(nn, level, sps)%viscSubface(mm)%q)), "gpuDownloadWallStress")
        end do
    end subroutine gpuDownloadWallStress

    ! RCCL bootstrap over the reference's own MPI communicator: rank 0 draws the unique id, it travels with the existing
    ! mpi_bcast, every rank joins.  Once, after adflow_gpu_init and before the first gpuRegisterComm with remote neighbours.
    subroutine gpuCommInit()
        use communication, only: adflow_comm_world, myID, nProc
        character(kind=c_char) :: id(128)
        integer :: ierr
        if (nProc == 1) return
        id = c_null_char
        if (myID == 0) call gpuCheck(adflow_gpu_comm_unique_id(id), "gpuCommInit")
        call mpi_bcast(id, 128, mpi_character, 0, adflow_comm_world, ierr)
        call gpuCheck(adflow_gpu_comm_init(int(myID, c_int), int(nProc, c_int), id), "gpuCommInit")
        ! every rank must have joined ONE communicator of nProc ranks under its MPI rank
        block
            integer(c_int) :: r, n, cnt, ur
            call gpuCheck(adflow_gpu_comm_info(r, n, cnt, ur), "gpuCommInit")
            if (cnt /= nProc .or. ur /= myID) call gpuCheck(1_c_int, "gpuCommInit: the RCCL communicator disagrees with MPI")
        end block
    end subroutine gpuCommInit

    ! flowDoms(nn,level,sps)%surfNodeIndices / %uv (determineWallAssociation, wallDistance.F90:1663-2002) -> device; once after
    ! computeWallDistance.  gpuUpdateWallDistances replaces the updateWallDistancesQuickly calls of a level (wallDistance.F90:36,
    ! blockette.F90:207-209) after updateXSurf filled wallDistanceData::xSurf.
    subroutine gpuRegisterWallAssociation(nn, level, sps)
        use block, only: flowDoms
        integer(kind=intType), intent(in) :: nn, level, sps
        if (.not. associated(flowDoms(nn, level, sps)%surfNodeIndices)) return
        call gpuCheck(adflow_gpu_wall_distance_register(int(nn, c_int), int(level, c_int), int(sps, c_int), &
                                                        c_loc(flowDoms(nn, level, sps)%surfNodeIndices), &
                                                        c_loc(flowDoms(nn, level, sps)%uv)), "gpuRegisterWallAssociation")
    end subroutine gpuRegisterWallAssociation

    subroutine gpuUpdateWallDistances(level)
        use wallDistanceData, only: xSurf
        integer(kind=intType), intent(in) :: level
        call gpuCheck(adflow_gpu_update_wall_distances(int(level, c_int), c_loc(xSurf), int(size(xSurf), c_int64_t)), &
                      "gpuUpdateWallDistances")
    end subroutine gpuUpdateWallDistances

end module adflowGpuShim
