! TEST INFRASTRUCTURE — NOT PRODUCT CODE.
!
! bind(C) driver around the reference's own per-block hot-path routines
! (compiled in place from /root/reference by oracle/refbuild/Makefile).  It
! aims the reference's module-global `blockPointers` at caller-owned buffers
! (numpy arrays, Fortran order, bounds exactly as the reference allocates them:
! SURVEY.md §8(a) row T) and calls the reference routine unchanged.  Nothing of
! the reference's arithmetic is restated here.
!
! Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
! the resulting oracle/_ref/libadflow_ref.so.
module ref_driver
    use iso_c_binding
    use constants
    implicit none


contains

    function cstr(s) result(f)
        character(kind=c_char), dimension(*), intent(in) :: s
        character(len=64) :: f
        integer :: i
        f = ' '
        do i = 1, 64
            if (s(i) == c_null_char) exit
            f(i:i) = s(i)
        end do
    end function cstr

    ! ------------------------------------------------------------------ dims
    subroutine ref_set_dims(nx_, ny_, nz_, nw_, nwf_) bind(C, name="ref_set_dims")
        use blockPointers
        use flowVarRefState, only: nw, nwf, nwt, nt1, nt2, wInf
        use cgnsGrid, only: massFlowFamilyInv, massFlowFamilyDiss
        use section, only: sections, nSections
        integer(c_int), value :: nx_, ny_, nz_, nw_, nwf_
        nx = nx_; ny = ny_; nz = nz_
        il = nx + 1; jl = ny + 1; kl = nz + 1
        ie = nx + 2; je = ny + 2; ke = nz + 2
        ib = nx + 3; jb = ny + 3; kb = nz + 3
        nw = nw_; nwf = nwf_; nwt = nw - nwf; nt1 = nwf + 1; nt2 = nw
        spectralSol = 1; sectionID = 1; nbkLocal = 1; nbkGlobal = 1
        rightHanded = .true.
        blockIsMoving = .false.; addGridVelocities = .false.
        nBocos = 0; nViscBocos = 0; nSubface = 0; n1to1 = 0
        if (.not. allocated(massFlowFamilyInv)) then
            allocate (massFlowFamilyInv(0:0, 1), massFlowFamilyDiss(0:0, 1))
        end if
        massFlowFamilyInv = zero; massFlowFamilyDiss = zero
        if (.not. allocated(wInf)) then
            allocate (wInf(10)); wInf = zero
        end if
        if (.not. allocated(sections)) then
            ! one steady, non-rotating, non-periodic section
            nSections = 1
            allocate (sections(1))
            sections(1)%periodic = .false.; sections(1)%rotating = .false.; sections(1)%nSlices = 1
            sections(1)%timePeriod = one; sections(1)%rotCenter = zero; sections(1)%translation = zero
            sections(1)%rotAxis = zero; sections(1)%rotRate = zero; sections(1)%rotMatrix = zero
        end if
    end subroutine ref_set_dims

    ! -------------------------------------------------------------- pointers
    subroutine ref_set_ptr(name, ptr) bind(C, name="ref_set_ptr")
        use blockPointers
        character(kind=c_char), dimension(*), intent(in) :: name
        type(c_ptr), value :: ptr
        real(kind=realType), dimension(:, :, :), pointer :: t3
        real(kind=realType), dimension(:, :, :, :), pointer :: t4
        integer(kind=intType), dimension(:, :, :), pointer :: i3
        integer(kind=intType), dimension(:, :), pointer :: i2
        real(kind=realType), dimension(:), pointer :: t1
        integer(kind=porType), dimension(:, :, :), pointer :: b3
        integer(kind=intType) :: nwl
        character(len=64) :: n
        n = cstr(name)
        select case (trim(n))
        case ('w', 'dw')
            nwl = ref_nw()
            call c_f_pointer(ptr, t4, [ib + 1, jb + 1, kb + 1, nwl])
            select case (trim(n))
            case ('w'); w(0:, 0:, 0:, 1:) => t4
            case ('dw'); dw(0:, 0:, 0:, 1:) => t4
            end select
        case ('wn', 'wr')
            call c_f_pointer(ptr, t4, [nx, ny, nz, ref_nwf()])
            select case (trim(n))
            case ('wn'); wn(2:, 2:, 2:, 1:) => t4
            case ('wr'); wr(2:, 2:, 2:, 1:) => t4
            end select
        case ('w1')
            call c_f_pointer(ptr, t4, [ie, je, ke, ref_nwf()])
            w1(1:, 1:, 1:, 1:) => t4
        case ('bmti1', 'bmti2')
            call c_f_pointer(ptr, t4, [je, ke, 1, 1])
            if (trim(n) == 'bmti1') bmti1(1:, 1:, ref_nt1():, ref_nt1():) => t4
            if (trim(n) == 'bmti2') bmti2(1:, 1:, ref_nt1():, ref_nt1():) => t4
        case ('bmtj1', 'bmtj2')
            call c_f_pointer(ptr, t4, [ie, ke, 1, 1])
            if (trim(n) == 'bmtj1') bmtj1(1:, 1:, ref_nt1():, ref_nt1():) => t4
            if (trim(n) == 'bmtj2') bmtj2(1:, 1:, ref_nt1():, ref_nt1():) => t4
        case ('bmtk1', 'bmtk2')
            call c_f_pointer(ptr, t4, [ie, je, 1, 1])
            if (trim(n) == 'bmtk1') bmtk1(1:, 1:, ref_nt1():, ref_nt1():) => t4
            if (trim(n) == 'bmtk2') bmtk2(1:, 1:, ref_nt1():, ref_nt1():) => t4
        case ('bvti1', 'bvti2')
            call c_f_pointer(ptr, t3, [je, ke, 1])
            if (trim(n) == 'bvti1') bvti1(1:, 1:, ref_nt1():) => t3
            if (trim(n) == 'bvti2') bvti2(1:, 1:, ref_nt1():) => t3
        case ('bvtj1', 'bvtj2')
            call c_f_pointer(ptr, t3, [ie, ke, 1])
            if (trim(n) == 'bvtj1') bvtj1(1:, 1:, ref_nt1():) => t3
            if (trim(n) == 'bvtj2') bvtj2(1:, 1:, ref_nt1():) => t3
        case ('bvtk1', 'bvtk2')
            call c_f_pointer(ptr, t3, [ie, je, 1])
            if (trim(n) == 'bvtk1') bvtk1(1:, 1:, ref_nt1():) => t3
            if (trim(n) == 'bvtk2') bvtk2(1:, 1:, ref_nt1():) => t3
        case ('fw')
            call c_f_pointer(ptr, t4, [ib + 1, jb + 1, kb + 1, ref_nwf()])
            fw(0:, 0:, 0:, 1:) => t4
        case ('scratch')
            call c_f_pointer(ptr, t4, [ib + 1, jb + 1, kb + 1, 10])
            scratch(0:, 0:, 0:, 1:) => t4
        case ('p', 'gamma', 'rlv', 'rev', 'aa', 'vol', 'volRef', 'shockSensor')
            call c_f_pointer(ptr, t3, [ib + 1, jb + 1, kb + 1])
            select case (trim(n))
            case ('p'); p(0:, 0:, 0:) => t3
            case ('gamma'); gamma(0:, 0:, 0:) => t3
            case ('rlv'); rlv(0:, 0:, 0:) => t3
            case ('rev'); rev(0:, 0:, 0:) => t3
            case ('aa'); aa(0:, 0:, 0:) => t3
            case ('vol'); vol(0:, 0:, 0:) => t3
            case ('volRef'); volRef(0:, 0:, 0:) => t3
            case ('shockSensor'); shockSensor(0:, 0:, 0:) => t3
            end select
        case ('iblank')
            call c_f_pointer(ptr, i3, [ib + 1, jb + 1, kb + 1])
            iblank(0:, 0:, 0:) => i3
        case ('indFamilyI', 'factFamilyI')
            call c_f_pointer(ptr, i3, [il, ny, nz])
            if (trim(n) == 'indFamilyI') indFamilyI(1:, 2:, 2:) => i3
            if (trim(n) == 'factFamilyI') factFamilyI(1:, 2:, 2:) => i3
        case ('indFamilyJ', 'factFamilyJ')
            call c_f_pointer(ptr, i3, [nx, jl, nz])
            if (trim(n) == 'indFamilyJ') indFamilyJ(2:, 1:, 2:) => i3
            if (trim(n) == 'factFamilyJ') factFamilyJ(2:, 1:, 2:) => i3
        case ('indFamilyK', 'factFamilyK')
            call c_f_pointer(ptr, i3, [nx, ny, kl])
            if (trim(n) == 'indFamilyK') indFamilyK(2:, 2:, 1:) => i3
            if (trim(n) == 'factFamilyK') factFamilyK(2:, 2:, 1:) => i3
        case ('viscIminPointer', 'viscImaxPointer')
            call c_f_pointer(ptr, i2, [ny, nz])
            if (trim(n) == 'viscIminPointer') viscIminPointer(2:, 2:) => i2
            if (trim(n) == 'viscImaxPointer') viscImaxPointer(2:, 2:) => i2
        case ('viscJminPointer', 'viscJmaxPointer')
            call c_f_pointer(ptr, i2, [nx, nz])
            if (trim(n) == 'viscJminPointer') viscJminPointer(2:, 2:) => i2
            if (trim(n) == 'viscJmaxPointer') viscJmaxPointer(2:, 2:) => i2
        case ('viscKminPointer', 'viscKmaxPointer')
            call c_f_pointer(ptr, i2, [nx, ny])
            if (trim(n) == 'viscKminPointer') viscKminPointer(2:, 2:) => i2
            if (trim(n) == 'viscKmaxPointer') viscKmaxPointer(2:, 2:) => i2
        case ('mgIFine')   ! (1:ie,2) on the COARSE block (coarseUtils.F90:254)
            call c_f_pointer(ptr, i2, [ie, 2]); mgIFine(1:, 1:) => i2
        case ('mgJFine')
            call c_f_pointer(ptr, i2, [je, 2]); mgJFine(1:, 1:) => i2
        case ('mgKFine')
            call c_f_pointer(ptr, i2, [ke, 2]); mgKFine(1:, 1:) => i2
        case ('mgICoarse')  ! (2:il,2) on the FINE block
            call c_f_pointer(ptr, i2, [nx, 2]); mgICoarse(2:, 1:) => i2
        case ('mgJCoarse')
            call c_f_pointer(ptr, i2, [ny, 2]); mgJCoarse(2:, 1:) => i2
        case ('mgKCoarse')
            call c_f_pointer(ptr, i2, [nz, 2]); mgKCoarse(2:, 1:) => i2
        case ('mgIWeight')
            call c_f_pointer(ptr, t1, [nx]); mgIWeight(2:) => t1
        case ('mgJWeight')
            call c_f_pointer(ptr, t1, [ny]); mgJWeight(2:) => t1
        case ('mgKWeight')
            call c_f_pointer(ptr, t1, [nz]); mgKWeight(2:) => t1
        case ('x')
            call c_f_pointer(ptr, t4, [ie + 1, je + 1, ke + 1, 3])
            x(0:, 0:, 0:, 1:) => t4
        case ('sFaceI')
            call c_f_pointer(ptr, t3, [ie + 1, je, ke]); sFaceI(0:, 1:, 1:) => t3
        case ('sFaceJ')
            call c_f_pointer(ptr, t3, [ie, je + 1, ke]); sFaceJ(1:, 0:, 1:) => t3
        case ('sFaceK')
            call c_f_pointer(ptr, t3, [ie, je, ke + 1]); sFaceK(1:, 1:, 0:) => t3
        case ('sI')
            call c_f_pointer(ptr, t4, [ie + 1, je, ke, 3])
            sI(0:, 1:, 1:, 1:) => t4
        case ('sJ')
            call c_f_pointer(ptr, t4, [ie, je + 1, ke, 3])
            sJ(1:, 0:, 1:, 1:) => t4
        case ('sK')
            call c_f_pointer(ptr, t4, [ie, je, ke + 1, 3])
            sK(1:, 1:, 0:, 1:) => t4
        case ('porI')
            call c_f_pointer(ptr, b3, [il, ny, nz])
            porI(1:, 2:, 2:) => b3
        case ('porJ')
            call c_f_pointer(ptr, b3, [nx, jl, nz])
            porJ(2:, 1:, 2:) => b3
        case ('porK')
            call c_f_pointer(ptr, b3, [nx, ny, kl])
            porK(2:, 2:, 1:) => b3
        case ('dtl', 'radI', 'radJ', 'radK')
            call c_f_pointer(ptr, t3, [ie, je, ke])
            select case (trim(n))
            case ('dtl'); dtl(1:, 1:, 1:) => t3
            case ('radI'); radI(1:, 1:, 1:) => t3
            case ('radJ'); radJ(1:, 1:, 1:) => t3
            case ('radK'); radK(1:, 1:, 1:) => t3
            end select
        case ('d2Wall')
            call c_f_pointer(ptr, t3, [nx, ny, nz])
            d2Wall(2:, 2:, 2:) => t3
        case ('pn')
            call c_f_pointer(ptr, t3, [nx, ny, nz])
            pn(2:, 2:, 2:) => t3
        case ('p1')
            call c_f_pointer(ptr, t3, [ie, je, ke])
            p1(1:, 1:, 1:) => t3
        case ('ux', 'uy', 'uz', 'vx', 'vy', 'vz', 'wx', 'wy', 'wz', 'qx', 'qy', 'qz')
            call c_f_pointer(ptr, t3, [il, jl, kl])
            select case (trim(n))
            case ('ux'); ux => t3
            case ('uy'); uy => t3
            case ('uz'); uz => t3
            case ('vx'); vx => t3
            case ('vy'); vy => t3
            case ('vz'); vz => t3
            case ('wx'); wx => t3
            case ('wy'); wy => t3
            case ('wz'); wz => t3
            case ('qx'); qx => t3
            case ('qy'); qy => t3
            case ('qz'); qz => t3
            end select
        case default
            print *, 'ref_set_ptr: unknown array ', trim(n)
            stop 1
        end select
    end subroutine ref_set_ptr

    integer function ref_nw()
        use flowVarRefState, only: nw
        ref_nw = nw
    end function ref_nw

    integer function ref_nwf()
        use flowVarRefState, only: nwf
        ref_nwf = nwf
    end function ref_nwf

    integer function ref_nt1()
        use flowVarRefState, only: nt1
        ref_nt1 = nt1
    end function ref_nt1

    ! --------------------------------------------------------------- scalars
    subroutine ref_set_int(name, v) bind(C, name="ref_set_int")
        use inputDiscretization
        use inputIteration
        use inputPhysics
        use iteration
        use flowVarRefState, only: viscous, eddyModel, kPresent
        use inputTimeSpectral, only: nTimeIntervalsSpectral
        use inputUnsteady, only: timeIntegrationScheme
        use blockPointers, only: rightHanded
        character(kind=c_char), dimension(*), intent(in) :: name
        integer(c_int), value :: v
        character(len=64) :: n
        n = cstr(name)
        select case (trim(n))
        case ('equations'); equations = v
        case ('rightHanded'); rightHanded = (v /= 0)
        case ('equationMode'); equationMode = v
        case ('spaceDiscr'); spaceDiscr = v
        case ('spaceDiscrCoarse'); spaceDiscrCoarse = v
        case ('limiter'); limiter = v
        case ('precond'); precond = v
        case ('orderTurb'); orderTurb = v
        case ('riemann'); riemann = v
        case ('riemannCoarse'); riemannCoarse = v
        case ('turbModel'); turbModel = v
        case ('turbProd'); turbProd = v
        case ('cpModel'); cpModel = v
        case ('smoother'); smoother = v
        case ('nRKStages'); nRKStages = v
        case ('rkStage'); rkStage = v
        case ('currentLevel'); currentLevel = v
        case ('groundLevel'); groundLevel = v
        case ('resAveraging'); resAveraging = v
        case ('turbTreatment'); turbTreatment = v
        case ('turbRelax'); turbRelax = v
        case ('eulerWallBCTreatment'); eulerWallBCTreatment = v
        case ('exchangePressureEarly'); exchangePressureEarly = (v /= 0)
        case ('viscWallBCTreatment'); viscWallBCTreatment = v
        case ('outflowTreatment'); outflowTreatment = v
        case ('nSubIterTurb'); nSubIterTurb = v
        case ('nSubiterations'); nSubiterations = v
        case ('nTimeIntervalsSpectral'); nTimeIntervalsSpectral = v
        case ('timeIntegrationScheme'); timeIntegrationScheme = v
        case ('viscous'); viscous = (v /= 0)
        case ('eddyModel'); eddyModel = (v /= 0)
        case ('kPresent'); kPresent = (v /= 0)
        case ('dirScaling'); dirScaling = (v /= 0)
        case ('lumpedDiss'); lumpedDiss = (v /= 0)
        case ('approxSA'); approxSA = (v /= 0)
        case ('radiiNeededFine'); radiiNeededFine = (v /= 0)
        case ('radiiNeededCoarse'); radiiNeededCoarse = (v /= 0)
        case ('lowSpeedPreconditioner'); lowSpeedPreconditioner = (v /= 0)
        case ('hScalingInlet'); hScalingInlet = (v /= 0)
        case ('useQCR'); useQCR = (v /= 0)
        case ('useRotationSA'); useRotationSA = (v /= 0)
        case ('useft2SA'); useft2SA = (v /= 0)
        case ('wallFunctions'); wallFunctions = (v /= 0)
        case ('useDissContinuation'); useDissContinuation = (v /= 0)
        case ('vortexCorr'); vortexCorr = (v /= 0)
        case default
            print *, 'ref_set_int: unknown name ', trim(n)
            stop 1
        end select
    end subroutine ref_set_int

    subroutine ref_set_real(name, v) bind(C, name="ref_set_real")
        use inputDiscretization
        use inputIteration
        use inputPhysics
        use iteration
        use flowVarRefState
        use paramTurb, only: rsaCw1
        character(kind=c_char), dimension(*), intent(in) :: name
        real(c_double), value :: v
        character(len=64) :: n
        n = cstr(name)
        select case (trim(n))
        case ('vis2'); vis2 = v
        case ('vis4'); vis4 = v
        case ('vis2Coarse'); vis2Coarse = v
        case ('adis'); adis = v
        case ('acousticScaleFactor'); acousticScaleFactor = v
        case ('kappaCoef'); kappaCoef = v
        case ('sigma'); sigma = v
        case ('cfl'); cfl = v
        case ('cflCoarse'); cflCoarse = v
        case ('cflLimit'); cflLimit = v
        case ('fcoll'); fcoll = v
        case ('smoop'); smoop = v
        case ('alfaTurb'); alfaTurb = v
        case ('betaTurb'); betaTurb = v
        case ('rFil'); rFil = v
        case ('totalR'); totalR = v
        case ('totalR0'); totalR0 = v
        case ('gammaConstant'); gammaConstant = v
        case ('gammaInf'); gammaInf = v
        case ('pInf'); pInf = v
        case ('pInfCorr'); pInfCorr = v
        case ('rhoInf'); rhoInf = v
        case ('uInf'); uInf = v
        case ('RGas'); RGas = v
        case ('muInf'); muInf = v
        case ('muRef'); muRef = v
        case ('TRef'); TRef = v
        case ('pRef'); pRef = v
        case ('uRef'); uRef = v
        case ('LRef'); LRef = v
        case ('ordersConverged'); ordersConverged = v
        case ('rhoRef'); rhoRef = v
        case ('timeRef'); timeRef = v
        case ('prandtl'); prandtl = v
        case ('prandtlTurb'); prandtlTurb = v
        case ('SSuthDim'); SSuthDim = v
        case ('muSuthDim'); muSuthDim = v
        case ('TSuthDim'); TSuthDim = v
        case ('SAKappa'); SAKappa = v
        case ('SAcb1'); SAcb1 = v
        case ('SAcb2'); SAcb2 = v
        case ('SAsigma'); SAsigma = v
        case ('SAcv1'); SAcv1 = v
        case ('SAcw1'); rsaCw1 = v
        case ('SAcw2'); SAcw2 = v
        case ('SAcw3'); SAcw3 = v
        case ('SAct1'); SAct1 = v
        case ('SAct2'); SAct2 = v
        case ('SAct3'); SAct3 = v
        case ('SAct4'); SAct4 = v
        case ('SAcrot'); SAcrot = v
        case ('eddyVisInfRatio'); eddyVisInfRatio = v
        case ('wallOffset'); wallOffset = v
        case ('pklim'); pklim = v
        case ('dissContMagnitude'); dissContMagnitude = v
        case ('dissContMidpoint'); dissContMidpoint = v
        case ('dissContSharpness'); dissContSharpness = v
        case default
            print *, 'ref_set_real: unknown name ', trim(n)
            stop 1
        end select
    end subroutine ref_set_real

    subroutine ref_set_vec(name, v, nv) bind(C, name="ref_set_vec")
        use inputIteration, only: etaRK, cdisRK, turbResScale
        use flowVarRefState, only: wInf
        character(kind=c_char), dimension(*), intent(in) :: name
        integer(c_int), value :: nv
        real(c_double), dimension(nv), intent(in) :: v
        character(len=64) :: n
        n = cstr(name)
        select case (trim(n))
        case ('etaRK')
            if (allocated(etaRK)) deallocate (etaRK)
            allocate (etaRK(nv)); etaRK = v
        case ('cdisRK')
            if (allocated(cdisRK)) deallocate (cdisRK)
            allocate (cdisRK(nv)); cdisRK = v
        case ('wInf')
            if (allocated(wInf)) deallocate (wInf)
            allocate (wInf(nv)); wInf = v
        case ('turbResScale')
            turbResScale(1:nv) = v
        case default
            print *, 'ref_set_vec: unknown name ', trim(n)
            stop 1
        end select
    end subroutine ref_set_vec

    subroutine ref_set_cycling(c, n) bind(C, name="ref_set_cycling")
        use iteration, only: cycling, nStepsCycling
        integer(c_int), value :: n
        integer(c_int), dimension(n), intent(in) :: c
        if (allocated(cycling)) deallocate (cycling)
        allocate (cycling(n))
        cycling = c
        nStepsCycling = n
    end subroutine ref_set_cycling

    ! --------------------------------------------------- boundary subfaces
    ! blockPointers%nBocos/BCType/BCFaceID/BCData(:)%{icBeg..jcEnd} for the block the
    ! pointers currently describe.  ranges(1:4, mm) = icBeg, icEnd, jcBeg, jcEnd.
    ! Fresh storage per call: committed blocks keep theirs.
    subroutine ref_set_bocos(nBocos_, nViscBocos_, types, faceIDs, ranges) bind(C, name="ref_set_bocos")
        use blockPointers
        integer(c_int), value :: nBocos_, nViscBocos_
        integer(c_int), intent(in) :: types(*), faceIDs(*), ranges(4, *)
        integer :: mm, n1, n2, iBeg, iEnd, jBeg, jEnd
        nBocos = nBocos_; nViscBocos = nViscBocos_
        nullify (BCType, BCFaceID, BCData, globalCell, s)
        allocate (BCType(max(nBocos, 1)), BCFaceID(max(nBocos, 1)), BCData(max(nBocos, 1)))
        allocate (globalCell(0:ib, 0:jb, 0:kb), s(0:ie, 0:je, 0:ke, 3))
        globalCell = 0; s = zero
        do mm = 1, nBocos
            BCType(mm) = types(mm); BCFaceID(mm) = faceIDs(mm)
            BCData(mm)%icBeg = ranges(1, mm); BCData(mm)%icEnd = ranges(2, mm)
            BCData(mm)%jcBeg = ranges(3, mm); BCData(mm)%jcEnd = ranges(4, mm)
            BCData(mm)%subsonicInletTreatment = 0
            nullify (BCData(mm)%norm, BCData(mm)%rface, BCData(mm)%uSlip, BCData(mm)%TNS_Wall, BCData(mm)%rho, &
                     BCData(mm)%velx, BCData(mm)%vely, BCData(mm)%velz, BCData(mm)%ps, BCData(mm)%ptInlet, &
                     BCData(mm)%ttInlet, BCData(mm)%htInlet, BCData(mm)%flowXdirInlet, BCData(mm)%flowYdirInlet, &
                     BCData(mm)%flowZdirInlet, BCData(mm)%turbInlet)
        end do
        ! node ranges of the subfaces: BCData%inBeg.. (face-local) and the block-level inBeg/jnBeg/knBeg arrays
        ! (blockPointers) that xhalo_block reads
        nSubface = nBocos
        nullify (inBeg, inEnd, jnBeg, jnEnd, knBeg, knEnd)
        allocate (inBeg(max(nBocos, 1)), inEnd(max(nBocos, 1)), jnBeg(max(nBocos, 1)), jnEnd(max(nBocos, 1)), &
                  knBeg(max(nBocos, 1)), knEnd(max(nBocos, 1)))
        do mm = 1, nBocos
            select case (BCFaceID(mm))
            case (iMin, iMax); n1 = jl; n2 = kl
            case (jMin, jMax); n1 = il; n2 = kl
            case default; n1 = il; n2 = jl
            end select
            iBeg = max(BCData(mm)%icBeg, 2); iEnd = min(BCData(mm)%icEnd, n1)
            jBeg = max(BCData(mm)%jcBeg, 2); jEnd = min(BCData(mm)%jcEnd, n2)
            BCData(mm)%inBeg = iBeg - 1; BCData(mm)%inEnd = iEnd
            BCData(mm)%jnBeg = jBeg - 1; BCData(mm)%jnEnd = jEnd
            BCData(mm)%symNorm = zero; BCData(mm)%symNormSet = .true.
            select case (BCFaceID(mm))
            case (iMin); inBeg(mm) = 1; inEnd(mm) = 1; jnBeg(mm) = iBeg - 1; jnEnd(mm) = iEnd; knBeg(mm) = jBeg - 1; knEnd(mm) = jEnd
            case (iMax); inBeg(mm) = il; inEnd(mm) = il; jnBeg(mm) = iBeg - 1; jnEnd(mm) = iEnd; knBeg(mm) = jBeg - 1; knEnd(mm) = jEnd
            case (jMin); jnBeg(mm) = 1; jnEnd(mm) = 1; inBeg(mm) = iBeg - 1; inEnd(mm) = iEnd; knBeg(mm) = jBeg - 1; knEnd(mm) = jEnd
            case (jMax); jnBeg(mm) = jl; jnEnd(mm) = jl; inBeg(mm) = iBeg - 1; inEnd(mm) = iEnd; knBeg(mm) = jBeg - 1; knEnd(mm) = jEnd
            case (kMin); knBeg(mm) = 1; knEnd(mm) = 1; inBeg(mm) = iBeg - 1; inEnd(mm) = iEnd; jnBeg(mm) = jBeg - 1; jnEnd(mm) = jEnd
            case (kMax); knBeg(mm) = kl; knEnd(mm) = kl; inBeg(mm) = iBeg - 1; inEnd(mm) = iEnd; jnBeg(mm) = jBeg - 1; jnEnd(mm) = jEnd
            end select
        end do
        ! what preprocessingAPI.F90:2430-2581 (viscSubfaceInfo) sets up: storage of the wall stress tensor / heat flux
        ! of the viscous subfaces (their owned face cells) and the visc*Pointer maps into it
        nullify (viscSubface)
        allocate (viscSubface(max(nViscBocos, 1)))
        if (associated(viscIminPointer)) then
            viscIminPointer = 0; viscImaxPointer = 0; viscJminPointer = 0
            viscJmaxPointer = 0; viscKminPointer = 0; viscKmaxPointer = 0
        end if
        do mm = 1, nViscBocos
            select case (BCFaceID(mm))
            case (iMin, iMax); n1 = jl; n2 = kl
            case (jMin, jMax); n1 = il; n2 = kl
            case default; n1 = il; n2 = jl
            end select
            iBeg = max(BCData(mm)%icBeg, 2); iEnd = min(BCData(mm)%icEnd, n1)
            jBeg = max(BCData(mm)%jcBeg, 2); jEnd = min(BCData(mm)%jcEnd, n2)
            BCData(mm)%inBeg = iBeg - 1; BCData(mm)%inEnd = iEnd
            BCData(mm)%jnBeg = jBeg - 1; BCData(mm)%jnEnd = jEnd
            allocate (viscSubface(mm)%tau(iBeg:iEnd, jBeg:jEnd, 6), viscSubface(mm)%q(iBeg:iEnd, jBeg:jEnd, 3), &
                      viscSubface(mm)%utau(iBeg:iEnd, jBeg:jEnd))
            viscSubface(mm)%tau = zero; viscSubface(mm)%q = zero; viscSubface(mm)%utau = zero
            select case (BCFaceID(mm))
            case (iMin); viscIminPointer(iBeg:iEnd, jBeg:jEnd) = mm
            case (iMax); viscImaxPointer(iBeg:iEnd, jBeg:jEnd) = mm
            case (jMin); viscJminPointer(iBeg:iEnd, jBeg:jEnd) = mm
            case (jMax); viscJmaxPointer(iBeg:iEnd, jBeg:jEnd) = mm
            case (kMin); viscKminPointer(iBeg:iEnd, jBeg:jEnd) = mm
            case (kMax); viscKmaxPointer(iBeg:iEnd, jBeg:jEnd) = mm
            end select
        end do
    end subroutine ref_set_bocos

    ! viscSubface(mm)%tau / %q of the current block -> caller arrays (n1, n2, 6) / (n1, n2, 3); dims returns n1, n2
    subroutine ref_get_wall_stress(mm, tau, q, dims) bind(C, name="ref_get_wall_stress")
        use blockPointers
        integer(c_int), value :: mm
        real(c_double), intent(out) :: tau(*), q(*)
        integer(c_int), intent(out) :: dims(2)
        integer :: n
        dims(1) = size(viscSubface(mm)%tau, 1); dims(2) = size(viscSubface(mm)%tau, 2)
        n = dims(1) * dims(2)
        tau(1:6 * n) = reshape(viscSubface(mm)%tau, [6 * n])
        q(1:3 * n) = reshape(viscSubface(mm)%q, [3 * n])
    end subroutine ref_get_wall_stress

    ! blockPointers%addGridVelocities / %blockIsMoving and cgnsDoms(nbkGlobal)%rotRate of the bound block
    subroutine ref_set_moving(addGridVel, isMoving, rotRate) bind(C, name="ref_set_moving")
        use blockPointers
        use cgnsGrid, only: cgnsDoms, cgnsNDom
        integer(c_int), value :: addGridVel, isMoving
        real(c_double), intent(in) :: rotRate(3)
        addGridVelocities = (addGridVel /= 0); blockIsMoving = (isMoving /= 0)
        if (.not. allocated(cgnsDoms)) then
            allocate (cgnsDoms(1)); cgnsNDom = 1
        end if
        cgnsDoms(1)%rotRate = rotRate; cgnsDoms(1)%rotCenter = zero; cgnsDoms(1)%rotatingFrameSpecified = (isMoving /= 0)
    end subroutine ref_set_moving

    ! actuatorRegions(iRegion) for the single bound block (blkPtr(0:1) = 0, n)
    subroutine ref_set_actuator(iRegion, nRegions, n, cellIDs, force, heat, volume, relaxStart, relaxEnd) &
        bind(C, name="ref_set_actuator")
        use actuatorRegionData
        integer(c_int), value :: iRegion, nRegions, n
        integer(c_int), intent(in) :: cellIDs(3, n)
        real(c_double), intent(in) :: force(3)
        real(c_double), value :: heat, volume, relaxStart, relaxEnd
        nActuatorRegions = nRegions
        if (nRegions == 0) return
        associate (r => actuatorRegions(iRegion))
            r%nCellIDs = n
            allocate (r%cellIDs(3, max(n, 1)))
            r%cellIDs(:, 1:n) = cellIDs
            if (allocated(r%blkPtr)) deallocate (r%blkPtr)
            allocate (r%blkPtr(0:1)); r%blkPtr(0) = 0; r%blkPtr(1) = n
            r%force = force; r%heat = heat; r%volume = volume; r%relaxStart = relaxStart; r%relaxEnd = relaxEnd
        end associate
    end subroutine ref_set_actuator

    subroutine ref_set_sym_norm(mm, v) bind(C, name="ref_set_sym_norm")
        use blockPointers
        integer(c_int), value :: mm
        real(c_double), intent(in) :: v(3)
        BCData(mm)%symNorm = v; BCData(mm)%symNormSet = .true.
    end subroutine ref_set_sym_norm

    subroutine ref_set_inlet_treatment(mm, v) bind(C, name="ref_set_inlet_treatment")
        use blockPointers
        integer(c_int), value :: mm, v
        BCData(mm)%subsonicInletTreatment = v
    end subroutine ref_set_inlet_treatment

    ! member `name` of BCData(mm) => caller-owned array with the reference's bounds
    subroutine ref_set_bcdata(mm, name, ptr) bind(C, name="ref_set_bcdata")
        use blockPointers
        use flowVarRefState, only: nt1, nt2
        integer(c_int), value :: mm
        character(kind=c_char), dimension(*), intent(in) :: name
        type(c_ptr), value :: ptr
        real(kind=realType), dimension(:, :, :), pointer :: t3
        real(kind=realType), dimension(:, :), pointer :: t2
        integer(kind=intType) :: i0, i1, j0, j1
        character(len=64) :: n
        n = cstr(name)
        i0 = BCData(mm)%icBeg; i1 = BCData(mm)%icEnd; j0 = BCData(mm)%jcBeg; j1 = BCData(mm)%jcEnd
        select case (trim(n))
        case ('norm', 'uSlip')
            call c_f_pointer(ptr, t3, [i1 - i0 + 1, j1 - j0 + 1, 3])
            if (trim(n) == 'norm') BCData(mm)%norm(i0:, j0:, 1:) => t3
            if (trim(n) == 'uSlip') BCData(mm)%uSlip(i0:, j0:, 1:) => t3
        case ('turbInlet')
            call c_f_pointer(ptr, t3, [i1 - i0 + 1, j1 - j0 + 1, nt2 - nt1 + 1])
            BCData(mm)%turbInlet(i0:, j0:, nt1:) => t3
        case ('rface', 'TNS_Wall', 'rho', 'velx', 'vely', 'velz', 'ps', 'ptInlet', 'ttInlet', 'htInlet', 'flowXdirInlet', &
              'flowYdirInlet', 'flowZdirInlet')
            call c_f_pointer(ptr, t2, [i1 - i0 + 1, j1 - j0 + 1])
            select case (trim(n))
            case ('ptInlet'); BCData(mm)%ptInlet(i0:, j0:) => t2
            case ('ttInlet'); BCData(mm)%ttInlet(i0:, j0:) => t2
            case ('htInlet'); BCData(mm)%htInlet(i0:, j0:) => t2
            case ('flowXdirInlet'); BCData(mm)%flowXdirInlet(i0:, j0:) => t2
            case ('flowYdirInlet'); BCData(mm)%flowYdirInlet(i0:, j0:) => t2
            case ('flowZdirInlet'); BCData(mm)%flowZdirInlet(i0:, j0:) => t2
            case ('rface'); BCData(mm)%rface(i0:, j0:) => t2
            case ('TNS_Wall'); BCData(mm)%TNS_Wall(i0:, j0:) => t2
            case ('rho'); BCData(mm)%rho(i0:, j0:) => t2
            case ('velx'); BCData(mm)%velx(i0:, j0:) => t2
            case ('vely'); BCData(mm)%vely(i0:, j0:) => t2
            case ('velz'); BCData(mm)%velz(i0:, j0:) => t2
            case ('ps'); BCData(mm)%ps(i0:, j0:) => t2
            end select
        case default
            print *, 'ref_set_bcdata: unknown member ', trim(n)
            stop 1
        end select
    end subroutine ref_set_bcdata

    ! ----------------------------------------------------------------- calls
    ! Each entry calls ONE reference routine, unchanged, on the current block.
    subroutine ref_call(name, iarg) bind(C, name="ref_call")
        use blockPointers
        use flowVarRefState, only: nw, nwf, nt1, nt2
        use solverUtils, only: timeStep_block
        use fluxes
        use residuals, only: residual_block, initres_block, computedwDADI, residualAveraging, sourceTerms_block
        use flowUtils, only: computeSpeedOfSoundSquared, allNodalGradients, computeEtotBlock, &
                             computePressureSimple, computeLamViscosity
        use turbUtils, only: computeEddyViscosity
        use sa, only: sa_block
        use adjointExtra, only: volume_block, metric_block, boundaryNormals, sumDwAndFw, xhalo_block
        use BCRoutines, only: applyAllBC_block
        use turbBCRoutines, only: bcTurbTreatment, applyAllTurbBCThisBlock
        character(kind=c_char), dimension(*), intent(in) :: name
        integer(c_int), value :: iarg
        character(len=64) :: n
        real(kind=realType) :: dummyReal
        n = cstr(name)
        select case (trim(n))
        case ('timeStep_block'); call timeStep_block(iarg /= 0)          ! solverUtils.F90:43
        case ('initres_flow'); call initres_block(1_intType, nwf, 1_intType, 1_intType)   ! residuals.F90:427
        case ('initres_all'); call initres_block(1_intType, nw, 1_intType, 1_intType)
        case ('initres_turb'); call initres_block(nt1, nt2, 1_intType, 1_intType)
        case ('inviscidCentralFlux'); call inviscidCentralFlux             ! fluxes.F90:4
        case ('inviscidDissFluxScalar'); call inviscidDissFluxScalar       ! fluxes.F90:1049
        case ('inviscidDissFluxMatrix'); call inviscidDissFluxMatrix       ! fluxes.F90:403
        case ('inviscidUpwindFlux'); call inviscidUpwindFlux(iarg /= 0)    ! fluxes.F90:1438
        case ('inviscidDissFluxScalarCoarse'); call inviscidDissFluxScalarCoarse
        case ('inviscidDissFluxMatrixCoarse'); call inviscidDissFluxMatrixCoarse
        case ('computeSpeedOfSoundSquared'); call computeSpeedOfSoundSquared
        case ('allNodalGradients'); call allNodalGradients
        case ('viscousFlux'); call viscousFlux                             ! fluxes.F90:2534
        case ('residual_block'); call residual_block                       ! residuals.F90:4
        case ('sumDwAndFw'); call sumDwAndFw
        case ('sa_block'); call sa_block(iarg /= 0)                        ! sa.F90:16
        case ('computedwDADI'); call computedwDADI                         ! residuals.F90:1062
        case ('residualAveraging'); call residualAveraging                 ! residuals.F90:1785
        case ('computeEtotBlock'); call computeEtotBlock(2_intType, il, 2_intType, jl, 2_intType, kl, iarg /= 0)
        case ('computePressureSimple'); call computePressureSimple(iarg /= 0)
        case ('computeLamViscosity'); call computeLamViscosity(iarg /= 0)
        case ('computeEddyViscosity'); call computeEddyViscosity(iarg /= 0)
        case ('volume_block'); call volume_block
        case ('metric_block'); call metric_block
        case ('boundaryNormals'); call boundaryNormals                      ! adjointExtra.F90:270
        case ('xhalo_block'); call xhalo_block                              ! adjointExtra.F90:365
        case ('sourceTerms_block'); call sourceTerms_block(1_intType, .true., int(iarg, intType), dummyReal)   ! residuals.F90:348
        case ('applyAllBC_block'); call applyAllBC_block(iarg /= 0)         ! BCRoutines.F90:57
        case ('bcTurbTreatment'); call bcTurbTreatment                       ! turbBCRoutines.F90:662
        case ('applyAllTurbBCThisBlock'); call applyAllTurbBCThisBlock(iarg /= 0)   ! turbBCRoutines.F90:49
        case ('zero_fw'); fw = zero
        case default
            print *, 'ref_call: unknown routine ', trim(n)
            stop 1
        end select
    end subroutine ref_call

    ! blockette::blockResCore (blockette.F90:755-852) cannot be linked here
    ! (module blockette pulls in haloExchange/BC data/PETSc); this entry
    ! issues the same sequence of reference calls in the same order.
    subroutine ref_block_res_core(updateIntermed, flowRes, turbRes) bind(C, name="ref_block_res_core")
        use blockPointers
        use flowVarRefState, only: nw, nwf, nt1, nt2, viscous
        use inputPhysics, only: equations, turbModel
        use inputDiscretization, only: spaceDiscr
        use solverUtils, only: timeStep_block
        use fluxes
        use residuals, only: initres_block
        use flowUtils, only: computeSpeedOfSoundSquared, allNodalGradients
        use sa, only: sa_block
        use adjointExtra, only: sumDwAndFw
        integer(c_int), value :: updateIntermed, flowRes, turbRes
        integer(kind=intType) :: lStart, lEnd
        lStart = 1; lEnd = nw
        if (flowRes /= 0 .and. turbRes == 0) then
            lEnd = nwf
        else if (flowRes == 0 .and. turbRes /= 0) then
            lStart = nt1; lEnd = nt2
        end if
        call timeStep_block(updateIntermed == 0)
        call initres_block(lStart, lEnd, 1_intType, 1_intType)
        fw = zero
        if (equations == RANSEquations .and. turbRes /= 0) then
            if (turbModel == spalartAllmaras) call sa_block(.true.)
        end if
        if (flowRes /= 0) then
            call inviscidCentralFlux
            select case (spaceDiscr)
            case (dissScalar); call inviscidDissFluxScalar
            case (dissMatrix); call inviscidDissFluxMatrix
            case (upwind); call inviscidUpwindFlux(.true.)
            end select
            if (viscous) then
                call computeSpeedOfSoundSquared
                call allNodalGradients
                call viscousFlux
            end if
            call sumDwAndFw
        end if
    end subroutine ref_block_res_core

    ! blockette::blocketteResCore (blockette.F90:299-753), the reference's DEFAULT residual path (useBlockettes = True,
    ! pyADflow.py:5734), called unchanged: metrics recomputed from x per 8^3 tile, fused SA routines, its own time step.
    ! blocketteRes sets rFil = one before the core (blockette.F90:270).
    subroutine ref_blockette_res_core(updateIntermed, flowRes, turbRes, dissApprox, viscApprox) &
        bind(C, name="ref_blockette_res_core")
        use blockette, only: blocketteResCore
        use iteration, only: rFil
        integer(c_int), value :: updateIntermed, flowRes, turbRes, dissApprox, viscApprox
        rFil = one
        call blocketteResCore(dissApprox /= 0, viscApprox /= 0, updateIntermed /= 0, flowRes /= 0, turbRes /= 0, .true.)
    end subroutine ref_blockette_res_core

    ! EXECUTES the ISO_C_BINDING host side of the drop-in boundary (adflow_amd/fortran/adflow_gpu_shim.F90) against the C-ABI
    ! library that is loaded in this process: options from the reference's option modules, every block of the level from
    ! flowDoms(nn,level,1) by c_loc, the boundary subfaces, the 1-to-1 patterns of communication.F90, then ONE blocketteRes
    ! evaluation (flags of include/adflow_gpu.h) and the residual back into flowDoms(nn,1,1)%dw -- what the call-site edits of
    ! INTEGRATION.md do inside blockette::blocketteRes.  nSmooth > 0 additionally runs that many smoother sweeps
    ! (RungeKuttaSmoother / DADISmoother bodies) and brings the state back.
    subroutine ref_shim_roundtrip(level, flags, withBocos, nSmooth) bind(C, name="ref_shim_roundtrip")
        use adflowGpuShim
        use block, only: nDom
        use communication, only: commPatternCell_1st, commPatternCell_2nd, internalCell_1st, internalCell_2nd
        use inputIteration, only: smoother
        integer(c_int), value :: level, flags, withBocos, nSmooth
        integer(kind=intType) :: nn, lev
        integer :: it
        lev = level
        call gpuCheck(adflow_gpu_release_all(), "ref_shim_roundtrip")
        call gpuRefreshOptions()
        do nn = 1, nDom
            call gpuRegisterBlock(nn, lev, 1_intType)
            call gpuCheck(adflow_gpu_upload_state(int(nn, c_int), level, 1_c_int), "upload_state")
            if (withBocos /= 0) call gpuRegisterBocos(nn, lev, 1_intType)
        end do
        call gpuRegisterComm(lev, 2_intType, commPatternCell_2nd(level), internalCell_2nd(level))
        call gpuRegisterComm(lev, 1_intType, commPatternCell_1st(level), internalCell_1st(level))
        call gpuCheck(adflow_gpu_block_res(level, flags), "block_res")
        do it = 1, nSmooth
            if (smoother == RungeKutta) then
                call gpuCheck(adflow_gpu_rk_smooth(level), "rk_smooth")
            else
                call gpuCheck(adflow_gpu_dadi_smooth(level), "dadi_smooth")
            end if
        end do
        do nn = 1, nDom
            call gpuCheck(adflow_gpu_download_residual(int(nn, c_int), level, 1_c_int), "download_residual")
            if (nSmooth > 0) call gpuCheck(adflow_gpu_download_state(int(nn, c_int), level, 1_c_int), "download_state")
        end do
    end subroutine ref_shim_roundtrip

    ! the same sequence with the approximate-residual switches of blockResCore (blockette.F90:755-852)
    subroutine ref_block_res_core2(updateIntermed, flowRes, turbRes, dissApprox, viscApprox) &
        bind(C, name="ref_block_res_core2")
        use blockPointers
        use flowVarRefState, only: nw, nwf, nt1, nt2, viscous
        use inputPhysics, only: equations, turbModel
        use inputDiscretization, only: spaceDiscr
        use solverUtils, only: timeStep_block
        use fluxes
        use residuals, only: initres_block
        use flowUtils, only: computeSpeedOfSoundSquared, allNodalGradients
        use sa, only: sa_block
        use adjointExtra, only: sumDwAndFw
        integer(c_int), value :: updateIntermed, flowRes, turbRes, dissApprox, viscApprox
        integer(kind=intType) :: lStart, lEnd
        lStart = 1; lEnd = nw
        if (flowRes /= 0 .and. turbRes == 0) then
            lEnd = nwf
        else if (flowRes == 0 .and. turbRes /= 0) then
            lStart = nt1; lEnd = nt2
        end if
        call timeStep_block(updateIntermed == 0)
        call initres_block(lStart, lEnd, 1_intType, 1_intType)
        fw = zero
        if (equations == RANSEquations .and. turbRes /= 0) then
            if (turbModel == spalartAllmaras) call sa_block(.true.)
        end if
        if (flowRes /= 0) then
            call inviscidCentralFlux
            if (dissApprox /= 0) then          ! blockette.F90:808-817
                select case (spaceDiscr)
                case (dissScalar); call inviscidDissFluxScalarApprox
                case (dissMatrix); call inviscidDissFluxMatrixApprox
                case (upwind); call inviscidUpwindFlux(.true.)
                end select
            else
                select case (spaceDiscr)
                case (dissScalar); call inviscidDissFluxScalar
                case (dissMatrix); call inviscidDissFluxMatrix
                case (upwind); call inviscidUpwindFlux(.true.)
                end select
            end if
            if (viscous) then
                call computeSpeedOfSoundSquared
                if (viscApprox /= 0) then
                    call viscousFluxApprox
                else
                    call allNodalGradients
                    call viscousFlux
                end if
            end if
            call sumDwAndFw
        end if
    end subroutine ref_block_res_core2


    ! wallDistance::updateWallDistancesQuickly (wallDistance.F90:36-120) on the current block.  Module wallDistance itself needs the
    ! ADT / overset search modules; the reference ships the same routine (primal + reverse) regenerated by Tapenade in
    ! adjoint/outputReverse/wallDistance_b.f90, which compiles on its own: that primal is called, unchanged.
    subroutine ref_update_wall_distances(ind, uvp, xs, n) bind(C, name="ref_update_wall_distances")
        use blockPointers
        use wallDistanceData, only: xSurf
        use walldistance_b, only: updateWallDistancesQuickly
        type(c_ptr), value :: ind, uvp, xs
        integer(c_int), value :: n
        integer(kind=intType), dimension(:, :, :, :), pointer :: ip
        real(kind=realType), dimension(:, :, :, :), pointer :: up
        call c_f_pointer(ind, ip, [4, nx, ny, nz])
        call c_f_pointer(uvp, up, [2, nx, ny, nz])
        flowDoms(1, 1, 1)%surfNodeIndices(1:, 2:, 2:, 2:) => ip
        flowDoms(1, 1, 1)%uv(1:, 2:, 2:, 2:) => up
        call c_f_pointer(xs, xSurf, [n])
        call updateWallDistancesQuickly(1_intType, 1_intType, 1_intType)
    end subroutine ref_update_wall_distances

    ! adjointUtils::setupStateResidualMatrix with useAD = F (adjointUtils.F90:7-715) on the CURRENT block, the PETSc calls replaced
    ! by stores into jac(nx, ny, nz, nState, nState, nStencil) [blk(ll, l) of stencil entry s at the row cell].  Module adjointUtils
    ! and masterRoutines cannot be compiled here (PETSc matrices, the AD routines), so the loop nest is restated around the
    ! reference's own routines: block_res_state (masterRoutines.F90:1214-1283) = closures with halos, turbulence and mean-flow
    ! boundary conditions, blocketteResCore / the blockResCore sequence, actuator sources, resScale; setFDReference /
    ! resetFDReference (:1971-2058), referenceShockSensor (:1909-1969), the colourings (:1089-1185), stencils.f90.
    subroutine ref_fd_jacobian(usePC, frozenTurb, turbOnly, viscPC, useBlockettes, delta, jac, nStateOut, nStencilOut) &
        bind(C, name="ref_fd_jacobian")
        use blockPointers
        use flowVarRefState, only: nw, nwf, nt1, nt2, viscous
        use inputPhysics, only: equations
        use inputDiscretization, only: lumpedDiss, acousticScaleFactor, orderTurb, spaceDiscr
        use iteration, only: rFil, currentLevel, groundLevel, rkStage
        use stencils
        use blockette, only: blocketteResCore
        use flowUtils, only: computePressureSimple, computeLamViscosity
        use turbUtils, only: computeEddyViscosity
        use BCRoutines, only: applyAllBC_block
        use turbBCRoutines, only: bcTurbTreatment, applyAllTurbBCThisBlock
        use residuals, only: sourceTerms_block
        use actuatorRegionData, only: nActuatorRegions
        use adjointExtra, only: resScale
        integer(c_int), value :: usePC, frozenTurb, turbOnly, viscPC, useBlockettes
        real(c_double), value :: delta
        type(c_ptr), value :: jac
        integer(c_int), intent(out) :: nStateOut, nStencilOut
        external :: initialize_stencils
        real(kind=realType), dimension(:, :, :, :, :, :), pointer :: Jm
        real(kind=realType), dimension(:, :, :, :), allocatable :: wtmp, dwtmp
        real(kind=realType), dimension(:, :, :, :, :), allocatable :: dw_deriv
        integer(kind=intType), dimension(:, :, :), allocatable :: color
        integer(kind=intType), dimension(:, :), pointer :: stencil
        integer(kind=intType) :: n_stencil, nColor, iColor, lStart, lEnd, nState, l, ll, i, j, k, ii, jj, kk, ist, orderTurbSave
        real(kind=realType) :: acousticScaleSave, one_over_dx
        logical :: flowRes, turbRes, resetToRANS

        call initialize_stencils
        if (turbOnly /= 0) then
            flowRes = .false.; turbRes = .true.; lStart = nt1; lEnd = nt2
        else if (frozenTurb /= 0) then
            flowRes = .true.; turbRes = .false.; lStart = 1; lEnd = nwf
        else
            flowRes = .true.; turbRes = .true.; lStart = 1; lEnd = nw
        end if
        nState = lEnd - lStart + 1
        rkStage = 0
        if (usePC /= 0) then
            if (viscous .and. viscPC /= 0) then
                stencil => visc_pc_stencil; n_stencil = N_visc_pc
            else
                stencil => euler_pc_stencil; n_stencil = N_euler_pc
            end if
            lumpedDiss = .true.
            acousticScaleSave = acousticScaleFactor
            acousticScaleFactor = one
            orderTurbSave = orderTurb
            orderTurb = firstOrder
        else
            if (viscous) then
                stencil => visc_drdw_stencil; n_stencil = N_visc_drdw
            else
                stencil => euler_drdw_stencil; n_stencil = N_euler_drdw
            end if
        end if
        nStateOut = int(nState, c_int)
        nStencilOut = int(n_stencil, c_int)
        call c_f_pointer(jac, Jm, [nx, ny, nz, nState, nState, n_stencil])
        Jm = zero
        one_over_dx = one / delta
        resetToRANS = .false.
        if (frozenTurb /= 0 .and. equations == RANSEquations) then
            equations = NSEquations
            resetToRANS = .true.
        end if
        allocate (wtmp(0:ib, 0:jb, 0:kb, nw), dwtmp(0:ib, 0:jb, 0:kb, nw), dw_deriv(2:il, 2:jl, 2:kl, nw, nw), color(0:ib, 0:jb, 0:kb))

        if (usePC /= 0) call shock_sensor
        ! setFDReference
        call res_state(.true., .true.)
        wtmp = w(0:ib, 0:jb, 0:kb, 1:nw)
        dwtmp = dw(0:ib, 0:jb, 0:kb, 1:nw)

        do k = 0, kb
            do j = 0, jb
                do i = 0, ib
                    if (usePC /= 0) then
                        if (viscous .and. viscPC /= 0) then
                            color(i, j, k) = mod(i, 3) + 3 * mod(j, 3) + 9 * mod(k, 3) + 1     ! setup_3x3x3_coloring
                        else
                            color(i, j, k) = mod(i + 5 * j + 4 * k, 7) + 1                     ! setup_PC_coloring
                        end if
                    else if (viscous) then
                        color(i, j, k) = mod(i + 19 * j + 11 * k, 35) + 1                      ! setup_dRdw_visc_coloring
                    else
                        color(i, j, k) = mod(i + 3 * j + 4 * k, 13) + 1                        ! setup_dRdw_euler_coloring
                    end if
                end do
            end do
        end do
        nColor = maxval(color)
        if (usePC /= 0 .and. .not. (viscous .and. viscPC /= 0)) nColor = 7
        if (usePC /= 0 .and. viscous .and. viscPC /= 0) nColor = 27
        if (usePC == 0 .and. viscous) nColor = 35
        if (usePC == 0 .and. .not. viscous) nColor = 13

        do iColor = 1, nColor
            dw_deriv = zero
            do l = lStart, lEnd
                w(0:ib, 0:jb, 0:kb, 1:nw) = wtmp
                do k = 0, kb
                    do j = 0, jb
                        do i = 0, ib
                            if (color(i, j, k) == iColor) w(i, j, k, l) = w(i, j, k, l) + delta
                        end do
                    end do
                end do
                call res_state(flowRes, turbRes)
                do ll = lStart, lEnd
                    dw_deriv(:, :, :, ll, l) = one_over_dx * (dw(2:il, 2:jl, 2:kl, ll) - dwtmp(2:il, 2:jl, 2:kl, ll))
                end do
            end do
            do k = 0, kb
                do j = 0, jb
                    do i = 0, ib
                        if (color(i, j, k) /= iColor) cycle
                        do ist = 1, n_stencil
                            ii = stencil(ist, 1); jj = stencil(ist, 2); kk = stencil(ist, 3)
                            if (i + ii >= 2 .and. i + ii <= il .and. j + jj >= 2 .and. j + jj <= jl .and. &
                                k + kk >= 2 .and. k + kk <= kl) then
                                Jm(i + ii - 1, j + jj - 1, k + kk - 1, :, :, ist) = &
                                    dw_deriv(i + ii, j + jj, k + kk, lStart:lEnd, lStart:lEnd)
                            end if
                        end do
                    end do
                end do
            end do
        end do

        ! resetFDReference and the switches back
        w(0:ib, 0:jb, 0:kb, 1:nw) = wtmp
        dw(0:ib, 0:jb, 0:kb, 1:nw) = dwtmp
        if (usePC /= 0) then
            lumpedDiss = .false.
            acousticScaleFactor = acousticScaleSave
            orderTurb = orderTurbSave
        end if
        if (resetToRANS) equations = RANSEquations
        deallocate (wtmp, dwtmp, dw_deriv, color)
    contains
        subroutine res_state(fRes, tRes)            ! masterRoutines.F90:1258-1280
            logical, intent(in) :: fRes, tRes
            integer(kind=intType) :: iRegion
            integer(c_int) :: da
            real(kind=realType) :: pLocal
            call computePressureSimple(.true.)
            call computeLamViscosity(.true.)
            call computeEddyViscosity(.true.)
            if (equations == RANSEquations) then
                call bcTurbTreatment
                call applyAllTurbBCThisBlock(.true.)
            end if
            call applyAllBC_block(.true.)
            rFil = one
            if (useBlockettes /= 0) then
                call blocketteResCore(lumpedDiss, lumpedDiss, .false., fRes, tRes, .true.)
            else
                da = merge(1_c_int, 0_c_int, lumpedDiss)
                call ref_block_res_core2(0_c_int, merge(1_c_int, 0_c_int, fRes), merge(1_c_int, 0_c_int, tRes), da, da)
            end if
            do iRegion = 1, nActuatorRegions
                call sourceTerms_block(1_intType, .true., iRegion, pLocal)
            end do
            call resScale
        end subroutine res_state

        subroutine shock_sensor                     ! adjointUtils.F90:1925-1966
            integer(kind=intType) :: i, j, k
            if (equations == EulerEquations .or. spaceDiscr == dissMatrix) then
                shockSensor(0:ib, 0:jb, 0:kb) = p(0:ib, 0:jb, 0:kb)
            else
                do k = 0, kb
                    do j = 2, jl
                        do i = 2, il
                            shockSensor(i, j, k) = p(i, j, k) / (w(i, j, k, irho)**gamma(i, j, k))
                        end do
                    end do
                end do
                do k = 2, kl
                    do j = 2, jl
                        do i = 0, ib
                            if (i > 1 .and. i < ie) cycle
                            shockSensor(i, j, k) = p(i, j, k) / (w(i, j, k, irho)**gamma(i, j, k))
                        end do
                    end do
                    do i = 2, il
                        do j = 0, jb
                            if (j > 1 .and. j < je) cycle
                            shockSensor(i, j, k) = p(i, j, k) / (w(i, j, k, irho)**gamma(i, j, k))
                        end do
                    end do
                end do
            end if
        end subroutine shock_sensor
    end subroutine ref_fd_jacobian


    ! adjointUtils::setupStateResidualMatrix with useAD = T (adjointUtils.F90:227-409) on flowDoms(1, 1, 1): the loop nest of the
    ! colours / state variables restated as in ref_fd_jacobian (the original stores through PETSc), the derivative storage of
    ! allocDerivativeValues / zeroADSeeds (adjointUtils.F90:717-1083) restated for one block; every arithmetic routine is the
    ! reference's OWN Tapenade output (src/adjoint/outputForward/*.f90) in the call sequence of masterRoutines::block_res_state_d
    ! (masterRoutines.F90:1285-1393).  Needs the block committed to flowDoms (ref_alloc_doms / ref_commit_block).
    subroutine ref_ad_jacobian(usePC, frozenTurb, turbOnly, viscPC, jac, nStateOut, nStencilOut) bind(C, name="ref_ad_jacobian")
        use block, only: flowDoms, flowDomsd
        use blockPointers
        use flowVarRefState
        use inputPhysics, only: equations, turbModel
        use inputDiscretization, only: lumpedDiss, acousticScaleFactor, orderTurb, spaceDiscr
        use inputAdjoint, only: viscPCopt => viscPC
        use iteration, only: rFil, currentLevel, groundLevel, rkStage
        use stencils
        use utils, only: setPointers_d, setPointers
        use flowutils_d, only: computePressureSimple_d, computeLamViscosity_d, computeSpeedOfSoundSquared_d, allNodalGradients_d
        use turbutils_d, only: computeEddyViscosity_d, turbAdvection_d
        use turbbcroutines_d, only: bcTurbTreatment_d, applyAllTurbBCThisBlock_d
        use BCExtra_d, only: applyAllBC_block_d
        use solverutils_d, only: timeStep_block_d
        use sa_d, only: saSource_d, saViscous_d, saResScale_d, qq
        use fluxes_d, only: inviscidCentralFlux_d, inviscidDissFluxScalar_d, inviscidDissFluxMatrix_d, inviscidUpwindFlux_d, &
                            inviscidDissFluxScalarApprox_d, inviscidDissFluxMatrixApprox_d, viscousFlux_d, viscousFluxApprox_d
        use adjointextra_d, only: sumDwAndFw_d, resScale_d
        integer(c_int), value :: usePC, frozenTurb, turbOnly, viscPC
        type(c_ptr), value :: jac
        integer(c_int), intent(out) :: nStateOut, nStencilOut
        external :: initialize_stencils
        real(kind=realType), dimension(:, :, :, :, :, :), pointer :: Jm
        real(kind=realType), dimension(:, :, :, :, :), allocatable :: dw_deriv
        integer(kind=intType), dimension(:, :, :), allocatable :: color
        integer(kind=intType), dimension(:, :), pointer :: stencil
        integer(kind=intType) :: n_stencil, nColor, iColor, lStart, lEnd, nState, l, ll, i, j, k, ii, jj, kk, ist, orderTurbSave, mm
        integer(kind=intType) :: iBeg, iStop, jBeg, jStop, qnBeg, qnStop, rnBeg, rnStop
        real(kind=realType) :: acousticScaleSave
        logical :: resetToRANS, viscPCSave

        call initialize_stencils
        if (turbOnly /= 0) then
            lStart = nt1; lEnd = nt2
        else if (frozenTurb /= 0) then
            lStart = 1; lEnd = nwf
        else
            lStart = 1; lEnd = nw
        end if
        nState = lEnd - lStart + 1
        rkStage = 0
        viscPCSave = viscPCopt
        viscPCopt = (viscPC /= 0)
        if (usePC /= 0) then
            if (viscous .and. viscPC /= 0) then
                stencil => visc_pc_stencil; n_stencil = N_visc_pc
            else
                stencil => euler_pc_stencil; n_stencil = N_euler_pc
            end if
            lumpedDiss = .true.
            acousticScaleSave = acousticScaleFactor
            acousticScaleFactor = one
            orderTurbSave = orderTurb
            orderTurb = firstOrder
        else
            if (viscous) then
                stencil => visc_drdw_stencil; n_stencil = N_visc_drdw
            else
                stencil => euler_drdw_stencil; n_stencil = N_euler_drdw
            end if
        end if
        nStateOut = int(nState, c_int)
        nStencilOut = int(n_stencil, c_int)
        call setPointers(1_intType, 1_intType, 1_intType)
        call c_f_pointer(jac, Jm, [nx, ny, nz, nState, nState, n_stencil])
        Jm = zero
        resetToRANS = .false.
        if (frozenTurb /= 0 .and. equations == RANSEquations) then
            equations = NSEquations
            resetToRANS = .true.
        end if

        ! ---- allocDerivativeValues + zeroADSeeds for the one block
        if (allocated(flowDomsd)) deallocate (flowDomsd)
        allocate (flowDomsd(1, 1, 1))
        if (allocated(winfd)) deallocate (winfd)
        allocate (winfd(size(winf)))
        winfd = zero
        rhoinfd = zero; uinfd = zero; pinfd = zero; pinfcorrd = zero; rgasd = zero; muinfd = zero; gammainfd = zero
        if (.not. associated(flowDoms(1, 1, 1)%d2wall)) allocate (flowDoms(1, 1, 1)%d2wall(2:il, 2:jl, 2:kl))
        associate (D => flowDomsd(1, 1, 1))
            allocate (D%x(0:ie, 0:je, 0:ke, 3), D%vol(0:ib, 0:jb, 0:kb), D%si(0:ie, 1:je, 1:ke, 3), D%sj(1:ie, 0:je, 1:ke, 3), &
                      D%sk(1:ie, 1:je, 0:ke, 3), D%rotMatrixI(il, 2:jl, 2:kl, 3, 3), D%rotMatrixJ(2:il, jl, 2:kl, 3, 3), &
                      D%rotMatrixK(2:il, 2:jl, kl, 3, 3), D%s(ie, je, ke, 3), D%sFaceI(0:ie, je, ke), D%sFaceJ(ie, 0:je, ke), &
                      D%sFaceK(ie, je, 0:ke), D%w(0:ib, 0:jb, 0:kb, 1:nw), D%dw(0:ib, 0:jb, 0:kb, 1:nw), D%fw(0:ib, 0:jb, 0:kb, 1:nw), &
                      D%scratch(0:ib, 0:jb, 0:kb, 5), D%p(0:ib, 0:jb, 0:kb), D%gamma(0:ib, 0:jb, 0:kb), D%aa(0:ib, 0:jb, 0:kb), &
                      D%ux(il, jl, kl), D%uy(il, jl, kl), D%uz(il, jl, kl), D%vx(il, jl, kl), D%vy(il, jl, kl), D%vz(il, jl, kl), &
                      D%wx(il, jl, kl), D%wy(il, jl, kl), D%wz(il, jl, kl), D%qx(il, jl, kl), D%qy(il, jl, kl), D%qz(il, jl, kl), &
                      D%rlv(0:ib, 0:jb, 0:kb), D%rev(0:ib, 0:jb, 0:kb), D%dtl(1:ie, 1:je, 1:ke), D%radI(1:ie, 1:je, 1:ke), &
                      D%radJ(1:ie, 1:je, 1:ke), D%radK(1:ie, 1:je, 1:ke), D%BCData(nBocos), &
                      D%bmti1(je, ke, nt1:nt2, nt1:nt2), D%bmti2(je, ke, nt1:nt2, nt1:nt2), D%bmtj1(ie, ke, nt1:nt2, nt1:nt2), &
                      D%bmtj2(ie, ke, nt1:nt2, nt1:nt2), D%bmtk1(ie, je, nt1:nt2, nt1:nt2), D%bmtk2(ie, je, nt1:nt2, nt1:nt2), &
                      D%bvti1(je, ke, nt1:nt2), D%bvti2(je, ke, nt1:nt2), D%bvtj1(ie, ke, nt1:nt2), D%bvtj2(ie, ke, nt1:nt2), &
                      D%bvtk1(ie, je, nt1:nt2), D%bvtk2(ie, je, nt1:nt2), D%d2Wall(2:il, 2:jl, 2:kl), D%viscSubface(nViscBocos))
            D%nBocos = nBocos; D%nViscBocos = nViscBocos
            D%x = zero; D%vol = zero; D%si = zero; D%sj = zero; D%sk = zero; D%rotMatrixI = zero; D%rotMatrixJ = zero
            D%rotMatrixK = zero; D%s = zero; D%sFaceI = zero; D%sFaceJ = zero; D%sFaceK = zero; D%w = zero; D%dw = zero; D%fw = zero
            D%scratch = zero; D%p = zero; D%gamma = zero; D%aa = zero; D%ux = zero; D%uy = zero; D%uz = zero; D%vx = zero; D%vy = zero
            D%vz = zero; D%wx = zero; D%wy = zero; D%wz = zero; D%qx = zero; D%qy = zero; D%qz = zero; D%rlv = zero; D%rev = zero
            D%dtl = zero; D%radI = zero; D%radJ = zero; D%radK = zero; D%bmti1 = zero; D%bmti2 = zero; D%bmtj1 = zero; D%bmtj2 = zero
            D%bmtk1 = zero; D%bmtk2 = zero; D%bvti1 = zero; D%bvti2 = zero; D%bvtj1 = zero; D%bvtj2 = zero; D%bvtk1 = zero
            D%bvtk2 = zero; D%d2Wall = zero
            do mm = 1, nBocos
                iBeg = BCData(mm)%icBeg; iStop = BCData(mm)%icEnd; jBeg = BCData(mm)%jcBeg; jStop = BCData(mm)%jcEnd
                qnBeg = BCData(mm)%inBeg; qnStop = BCData(mm)%inEnd; rnBeg = BCData(mm)%jnBeg; rnStop = BCData(mm)%jnEnd
                allocate (D%BCData(mm)%norm(iBeg:iStop, jBeg:jStop, 3), D%BCData(mm)%rface(iBeg:iStop, jBeg:jStop), &
                          D%BCData(mm)%Fp(qnBeg + 1:qnStop, rnBeg + 1:rnStop, 3), D%BCData(mm)%Fv(qnBeg + 1:qnStop, rnBeg + 1:rnStop, 3), &
                          D%BCData(mm)%Tp(qnBeg:qnStop, rnBeg:rnStop, 3), D%BCData(mm)%Tv(qnBeg:qnStop, rnBeg:rnStop, 3), &
                          D%BCData(mm)%F(qnBeg:qnStop, rnBeg:rnStop, 3), D%BCData(mm)%T(qnBeg:qnStop, rnBeg:rnStop, 3), &
                          D%BCData(mm)%area(qnBeg + 1:qnStop, rnBeg + 1:rnStop), D%BCData(mm)%uSlip(iBeg:iStop, jBeg:jStop, 3), &
                          D%BCData(mm)%TNS_Wall(iBeg:iStop, jBeg:jStop), D%BCData(mm)%ptInlet(iBeg:iStop, jBeg:jStop), &
                          D%BCData(mm)%htInlet(iBeg:iStop, jBeg:jStop), D%BCData(mm)%ttInlet(iBeg:iStop, jBeg:jStop), &
                          D%BCData(mm)%turbInlet(iBeg:iStop, jBeg:jStop, nt1:nt2), D%BCData(mm)%ps(iBeg:iStop, jBeg:jStop))
                D%BCData(mm)%norm = zero; D%BCData(mm)%rface = zero; D%BCData(mm)%Fp = zero; D%BCData(mm)%Fv = zero
                D%BCData(mm)%Tp = zero; D%BCData(mm)%Tv = zero; D%BCData(mm)%F = zero; D%BCData(mm)%T = zero; D%BCData(mm)%area = zero
                D%BCData(mm)%uSlip = zero; D%BCData(mm)%TNS_Wall = zero; D%BCData(mm)%ptInlet = zero; D%BCData(mm)%htInlet = zero
                D%BCData(mm)%ttInlet = zero; D%BCData(mm)%turbInlet = zero; D%BCData(mm)%ps = zero
            end do
            do mm = 1, nViscBocos
                iBeg = BCData(mm)%inBeg + 1; iStop = BCData(mm)%inEnd; jBeg = BCData(mm)%jnBeg + 1; jStop = BCData(mm)%jnEnd
                allocate (D%viscSubface(mm)%tau(iBeg:iStop, jBeg:jStop, 6), D%viscSubface(mm)%q(iBeg:iStop, jBeg:jStop, 6))
                D%viscSubface(mm)%tau = zero; D%viscSubface(mm)%q = zero
            end do
        end associate
        call setPointers_d(1_intType, 1_intType, 1_intType)

        allocate (dw_deriv(2:il, 2:jl, 2:kl, nw, nw), color(0:ib, 0:jb, 0:kb))
        if (usePC /= 0) call shock_sensor_ad
        do k = 0, kb
            do j = 0, jb
                do i = 0, ib
                    if (usePC /= 0) then
                        if (viscous .and. viscPC /= 0) then
                            color(i, j, k) = mod(i, 3) + 3 * mod(j, 3) + 9 * mod(k, 3) + 1
                        else
                            color(i, j, k) = mod(i + 5 * j + 4 * k, 7) + 1
                        end if
                    else if (viscous) then
                        color(i, j, k) = mod(i + 19 * j + 11 * k, 35) + 1
                    else
                        color(i, j, k) = mod(i + 3 * j + 4 * k, 13) + 1
                    end if
                end do
            end do
        end do
        if (usePC /= 0 .and. .not. (viscous .and. viscPC /= 0)) nColor = 7
        if (usePC /= 0 .and. viscous .and. viscPC /= 0) nColor = 27
        if (usePC == 0 .and. viscous) nColor = 35
        if (usePC == 0 .and. .not. viscous) nColor = 13

        if (equations == RANSEquations .and. .not. allocated(qq)) allocate (qq(2:il, 2:jl, 2:kl))
        do iColor = 1, nColor
            dw_deriv = zero
            do l = lStart, lEnd
                wd = zero
                do k = 0, kb
                    do j = 0, jb
                        do i = 0, ib
                            if (color(i, j, k) == iColor) wd(i, j, k, l) = one
                        end do
                    end do
                end do
                call res_state_d
                do ll = lStart, lEnd
                    dw_deriv(:, :, :, ll, l) = dwd(2:il, 2:jl, 2:kl, ll)
                end do
            end do
            do k = 0, kb
                do j = 0, jb
                    do i = 0, ib
                        if (color(i, j, k) /= iColor) cycle
                        do ist = 1, n_stencil
                            ii = stencil(ist, 1); jj = stencil(ist, 2); kk = stencil(ist, 3)
                            if (i + ii >= 2 .and. i + ii <= il .and. j + jj >= 2 .and. j + jj <= jl .and. &
                                k + kk >= 2 .and. k + kk <= kl) then
                                Jm(i + ii - 1, j + jj - 1, k + kk - 1, :, :, ist) = &
                                    dw_deriv(i + ii, j + jj, k + kk, lStart:lEnd, lStart:lEnd)
                            end if
                        end do
                    end do
                end do
            end do
        end do
        if (allocated(qq)) deallocate (qq)
        if (usePC /= 0) then
            lumpedDiss = .false.
            acousticScaleFactor = acousticScaleSave
            orderTurb = orderTurbSave
        end if
        viscPCopt = viscPCSave
        if (resetToRANS) equations = RANSEquations
        deallocate (dw_deriv, color)
    contains
        subroutine res_state_d                      ! masterRoutines.F90:1319-1392
            call computePressureSimple_d(.true.)
            call computeLamViscosity_d(.true.)
            call computeEddyViscosity_d(.true.)
            if (equations == RANSEquations) then
                call bcTurbTreatment_d
                call applyAllTurbBCThisBlock_d(.true.)
            end if
            call applyAllBC_block_d(.true.)
            rFil = one
            call timeStep_block_d(.false.)
            dw = zero
            dwd = zero
            if (equations == RANSEquations) then
                if (turbModel == spalartAllmaras) then
                    call saSource_d
                    call turbAdvection_d(1_intType, 1_intType, itu1 - 1, qq)
                    call saViscous_d
                    call saResScale_d
                end if
            end if
            call inviscidCentralFlux_d
            if (lumpedDiss) then
                select case (spaceDiscr)
                case (dissScalar); call inviscidDissFluxScalarApprox_d
                case (dissMatrix); call inviscidDissFluxMatrixApprox_d
                case (upwind); call inviscidUpwindFlux_d(.true.)
                end select
            else
                select case (spaceDiscr)
                case (dissScalar); call inviscidDissFluxScalar_d
                case (dissMatrix); call inviscidDissFluxMatrix_d
                case (upwind); call inviscidUpwindFlux_d(.true.)
                end select
            end if
            if (viscous) then
                call computeSpeedOfSoundSquared_d
                if (.not. lumpedDiss .or. viscPCopt) then
                    call allNodalGradients_d
                    call viscousFlux_d
                else
                    call viscousFluxApprox_d
                end if
            end if
            call sumDwAndFw_d
            call resScale_d
        end subroutine res_state_d

        subroutine shock_sensor_ad                  ! referenceShockSensor, adjointUtils.F90:1925-1966
            integer(kind=intType) :: i, j, k
            if (equations == EulerEquations .or. spaceDiscr == dissMatrix) then
                shockSensor(0:ib, 0:jb, 0:kb) = p(0:ib, 0:jb, 0:kb)
            else
                do k = 0, kb
                    do j = 2, jl
                        do i = 2, il
                            shockSensor(i, j, k) = p(i, j, k) / (w(i, j, k, irho)**gamma(i, j, k))
                        end do
                    end do
                end do
                do k = 2, kl
                    do j = 2, jl
                        do i = 0, ib
                            if (i > 1 .and. i < ie) cycle
                            shockSensor(i, j, k) = p(i, j, k) / (w(i, j, k, irho)**gamma(i, j, k))
                        end do
                    end do
                    do i = 2, il
                        do j = 0, jb
                            if (j > 1 .and. j < je) cycle
                            shockSensor(i, j, k) = p(i, j, k) / (w(i, j, k, irho)**gamma(i, j, k))
                        end do
                    end do
                end do
            end if
        end subroutine shock_sensor_ad
    end subroutine ref_ad_jacobian


    ! ===================================================================
    ! multi-block mode: the reference's SHELL routines (smoothers, halo
    ! exchange, multigrid) loop over flowDoms(nn,level,sps) and re-aim
    ! blockPointers with utils::setPointers (utils.F90:3236).  ref_commit_block
    ! stores the current blockPointers association into flowDoms(nn,level,1).
    ! ===================================================================
    subroutine ref_alloc_doms(nDom_, nLevels_) bind(C, name="ref_alloc_doms")
        use block, only: flowDoms, nDom
        use communication
        use inputTimeSpectral, only: nTimeIntervalsSpectral
        use inputIteration, only: nMGLevels
        integer(c_int), value :: nDom_, nLevels_
        integer :: l
        if (allocated(flowDoms)) deallocate (flowDoms)
        allocate (flowDoms(nDom_, nLevels_, 1))
        nDom = nDom_
        nTimeIntervalsSpectral = 1
        nMGLevels = nLevels_
        myID = 0; nProc = 1
        if (allocated(commPatternCell_1st)) deallocate (commPatternCell_1st, commPatternCell_2nd, &
                                                        internalCell_1st, internalCell_2nd)
        allocate (commPatternCell_1st(nLevels_), commPatternCell_2nd(nLevels_), &
                  internalCell_1st(nLevels_), internalCell_2nd(nLevels_))
        if (allocated(commPatternNode_1st)) deallocate (commPatternNode_1st, internalNode_1st)
        allocate (commPatternNode_1st(nLevels_), internalNode_1st(nLevels_))
        do l = 1, nLevels_
            commPatternNode_1st(l)%nProcSend = 0; commPatternNode_1st(l)%nProcRecv = 0; commPatternNode_1st(l)%nPeriodic = 0
            internalNode_1st(l)%ncopy = 0; internalNode_1st(l)%nPeriodic = 0
        end do
        if (allocated(commPatternOverset)) deallocate (commPatternOverset, internalOverset)
        allocate (commPatternOverset(nLevels_, 1), internalOverset(nLevels_, 1))
        do l = 1, nLevels_
            commPatternCell_1st(l)%nProcSend = 0; commPatternCell_1st(l)%nProcRecv = 0; commPatternCell_1st(l)%nPeriodic = 0
            commPatternCell_2nd(l)%nProcSend = 0; commPatternCell_2nd(l)%nProcRecv = 0; commPatternCell_2nd(l)%nPeriodic = 0
            internalCell_1st(l)%ncopy = 0; internalCell_1st(l)%nPeriodic = 0
            internalCell_2nd(l)%ncopy = 0; internalCell_2nd(l)%nPeriodic = 0
            commPatternOverset(l, 1)%nProcSend = 0; commPatternOverset(l, 1)%nProcRecv = 0
            commPatternOverset(l, 1)%nPeriodic = 0
            internalOverset(l, 1)%ncopy = 0; internalOverset(l, 1)%nPeriodic = 0
        end do
        if (.not. allocated(sendBuffer)) then
            allocate (sendBuffer(1), recvBuffer(1), sendRequests(1), recvRequests(1))
        end if
    end subroutine ref_alloc_doms

    subroutine ref_commit_block(nn, level) bind(C, name="ref_commit_block")
        use block, only: flowDoms
        use blockPointers
        integer(c_int), value :: nn, level
        associate (d => flowDoms(nn, level, 1))
            d%nx = nx; d%ny = ny; d%nz = nz
            d%il = il; d%jl = jl; d%kl = kl
            d%ie = ie; d%je = je; d%ke = ke
            d%ib = ib; d%jb = jb; d%kb = kb
            d%cgnsBlockID = 1
            d%rightHanded = rightHanded
            d%iBegor = 1; d%iEndor = il; d%jBegor = 1; d%jEndor = jl; d%kBegor = 1; d%kEndor = kl
            d%nSubface = nBocos; d%n1to1 = 0; d%nBocos = nBocos; d%nViscBocos = nViscBocos
            d%BCType => BCType; d%BCFaceID => BCFaceID; d%BCData => BCData
            d%globalCell => globalCell; d%s => s; d%viscSubface => viscSubface
            d%inBeg => inBeg; d%inEnd => inEnd; d%jnBeg => jnBeg; d%jnEnd => jnEnd; d%knBeg => knBeg; d%knEnd => knEnd
            d%nOrphans = 0
            d%blockIsMoving = blockIsMoving; d%addGridVelocities = addGridVelocities
            if (addGridVelocities) then
                d%sFaceI => sFaceI; d%sFaceJ => sFaceJ; d%sFaceK => sFaceK
            end if
            d%iblank => iblank
            d%viscIminPointer => viscIminPointer; d%viscImaxPointer => viscImaxPointer
            d%viscJminPointer => viscJminPointer; d%viscJmaxPointer => viscJmaxPointer
            d%viscKminPointer => viscKminPointer; d%viscKmaxPointer => viscKmaxPointer
            d%x => x; d%si => si; d%sj => sj; d%sk => sk; d%vol => vol; d%volRef => volRef
            d%porI => porI; d%porJ => porJ; d%porK => porK
            d%indFamilyI => indFamilyI; d%indFamilyJ => indFamilyJ; d%indFamilyK => indFamilyK
            d%factFamilyI => factFamilyI; d%factFamilyJ => factFamilyJ; d%factFamilyK => factFamilyK
            d%w => w; d%p => p; d%aa => aa; d%gamma => gamma; d%rlv => rlv; d%rev => rev
            d%ux => ux; d%uy => uy; d%uz => uz; d%vx => vx; d%vy => vy; d%vz => vz
            d%wx => wx; d%wy => wy; d%wz => wz; d%qx => qx; d%qy => qy; d%qz => qz
            d%dw => dw; d%fw => fw; d%scratch => scratch
            d%shockSensor => shockSensor
            d%p1 => p1; d%w1 => w1; d%wr => wr
            d%wn => wn; d%pn => pn; d%dtl => dtl; d%radI => radI; d%radJ => radJ; d%radK => radK
            d%d2Wall => d2Wall
            d%bmti1 => bmti1; d%bmti2 => bmti2; d%bmtj1 => bmtj1; d%bmtj2 => bmtj2; d%bmtk1 => bmtk1; d%bmtk2 => bmtk2
            d%bvti1 => bvti1; d%bvti2 => bvti2; d%bvtj1 => bvtj1; d%bvtj2 => bvtj2; d%bvtk1 => bvtk1; d%bvtk2 => bvtk2
            d%mgIFine => mgIFine; d%mgJFine => mgJFine; d%mgKFine => mgKFine
            d%mgIWeight => mgIWeight; d%mgJWeight => mgJWeight; d%mgKWeight => mgKWeight
            d%mgICoarse => mgICoarse; d%mgJCoarse => mgJCoarse; d%mgKCoarse => mgKCoarse
        end associate
    end subroutine ref_commit_block

    ! same-process 1-to-1 halo copy lists (communication.F90 internalCommType):
    ! indices are 0-based cell indices exactly as the reference stores them
    ! (the +1 offset is applied at use, haloExchange.F90:605-607)
    subroutine ref_set_internal_comm(level, nLayers, ncopy, donorBlock, donorIdx, haloBlock, haloIdx) &
        bind(C, name="ref_set_internal_comm")
        use communication
        integer(c_int), value :: level, nLayers, ncopy
        integer(c_int), dimension(ncopy), intent(in) :: donorBlock, haloBlock
        integer(c_int), dimension(ncopy, 3), intent(in) :: donorIdx, haloIdx
        if (nLayers == 0) then
            call fill(internalNode_1st(level))      ! node pattern of exchangeCoor
        else if (nLayers == 1) then
            call fill(internalCell_1st(level))
        else
            call fill(internalCell_2nd(level))
        end if
    contains
        subroutine fill(ic)
            type(internalCommType), intent(inout) :: ic
            ic%ncopy = ncopy
            ic%nPeriodic = 0
            allocate (ic%donorBlock(ncopy), ic%haloBlock(ncopy), ic%donorIndices(ncopy, 3), ic%haloIndices(ncopy, 3))
            ic%donorBlock = donorBlock; ic%haloBlock = haloBlock
            ic%donorIndices = donorIdx; ic%haloIndices = haloIdx
        end subroutine fill
    end subroutine ref_set_internal_comm

    ! periodicData(nn) of internalNode_1st / internalCell_1st / internalCell_2nd(level) (nLayers = 0 / 1 / 2); the first call
    ! (nn = 1) sizes the list
    subroutine ref_set_periodic(level, nLayers, nn, nPeriodic, rotMatrix, rotCenter, translation, nHalos, blk, idx) &
        bind(C, name="ref_set_periodic")
        use communication
        integer(c_int), value :: level, nLayers, nn, nPeriodic, nHalos
        real(c_double), intent(in) :: rotMatrix(3, 3), rotCenter(3), translation(3)
        integer(c_int), intent(in) :: blk(nHalos), idx(nHalos, 3)
        if (nLayers == 0) then
            call fill(internalNode_1st(level))
        else if (nLayers == 1) then
            call fill(internalCell_1st(level))
        else
            call fill(internalCell_2nd(level))
        end if
    contains
        subroutine fill(ic)
            type(internalCommType), intent(inout) :: ic
            if (nn == 1) then
                ic%nPeriodic = nPeriodic
                allocate (ic%periodicData(max(nPeriodic, 1)))
            end if
            if (nPeriodic == 0) return
            ic%periodicData(nn)%rotMatrix = rotMatrix
            ic%periodicData(nn)%rotCenter = rotCenter
            ic%periodicData(nn)%translation = translation
            ic%periodicData(nn)%nHalos = nHalos
            allocate (ic%periodicData(nn)%block(nHalos), ic%periodicData(nn)%indices(nHalos, 3))
            ic%periodicData(nn)%block = blk
            ic%periodicData(nn)%indices = idx
        end subroutine fill
    end subroutine ref_set_periodic

    ! shell routines acting on every committed block of `level`
    subroutine ref_call_level(name, level, i1, i2) bind(C, name="ref_call_level")
        use iteration, only: currentLevel, groundLevel, rkStage
        use flowVarRefState, only: nwf, nw, nt1, nt2
        use haloExchange, only: whalo1, whalo2, exchangeCoor
        use smoothers, only: RungeKuttaSmoother, DADISmoother
        use solverUtils, only: timeStep
        use residuals, only: initres, residual
        use multiGrid, only: transferToCoarseGrid, transferToFineGrid, executeMGCycle
        use utils, only: setPointers
        use BCRoutines, only: applyAllBC
        character(kind=c_char), dimension(*), intent(in) :: name
        integer(c_int), value :: level, i1, i2
        character(len=64) :: n
        n = cstr(name)
        currentLevel = level
        select case (trim(n))
        case ('setPointers'); call setPointers(i1, level, 1_intType)
        case ('whalo2'); call whalo2(level, i1, i2, .true., .true., .true.)         ! haloExchange.F90:109
        case ('whalo1'); call whalo1(level, i1, i2, .true., .true., .true.)         ! haloExchange.F90:5
        case ('whalo2_turb'); call whalo2(level, i1, i2, .false., .false., .true.)   ! turbAPI.F90:92
        case ('timeStep'); call timeStep(i1 /= 0)                                     ! solverUtils.F90:4
        case ('initres'); call initres(i1, i2)                                        ! residuals.F90:964
        case ('residual'); call residual                                              ! residuals.F90:1028
        case ('RungeKuttaSmoother'); call RungeKuttaSmoother                          ! smoothers.F90:4
        case ('DADISmoother'); call DADISmoother                                      ! smoothers.F90:383
        case ('transferToCoarseGrid'); call transferToCoarseGrid                      ! multiGrid.F90:5
        case ('transferToFineGrid'); call transferToFineGrid(i1 /= 0)                 ! multiGrid.F90:326
        case ('executeMGCycle'); call executeMGCycle                                  ! multiGrid.F90:825
        case ('applyAllBC'); call applyAllBC(i1 /= 0)                                 ! BCRoutines.F90:15
        case ('exchangeCoor'); call exchangeCoor(level)                               ! haloExchange.F90:2456
        case default
            print *, 'ref_call_level: unknown routine ', trim(n)
            stop 1
        end select
    end subroutine ref_call_level


    ! sizes of the ISO_C_BINDING mirrors in adflow_amd/fortran/adflow_gpu_shim.F90
    ! (compiled here against the reference's modules) for the ABI cross-check
    subroutine ref_shim_sizes(opts_bytes, desc_bytes) bind(C, name="ref_shim_sizes")
        use adflowGpuShim, only: adflow_opts, adflow_block_desc
        integer(c_int), intent(out) :: opts_bytes, desc_bytes
        type(adflow_opts) :: o
        type(adflow_block_desc) :: d
        opts_bytes = int(c_sizeof(o), c_int)
        desc_bytes = int(c_sizeof(d), c_int)
    end subroutine ref_shim_sizes

    subroutine ref_shim_sizes2(bc_bytes, comm_bytes) bind(C, name="ref_shim_sizes2")
        use adflowGpuShim, only: adflow_bc_subface, adflow_comm_pattern
        integer(c_int), intent(out) :: bc_bytes, comm_bytes
        type(adflow_bc_subface) :: f
        type(adflow_comm_pattern) :: p
        bc_bytes = int(c_sizeof(f), c_int)
        comm_bytes = int(c_sizeof(p), c_int)
    end subroutine ref_shim_sizes2

end module ref_driver
