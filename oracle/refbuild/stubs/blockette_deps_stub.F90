! TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
! Name-only stand-ins for the modules NKSolver/blockette.F90 `use`s in its DRIVER routine blocketteRes (BC data update,
! wall-distance update, overset connectivity, surface integration: out of the hot path and not compilable here without
! CGNS / PETSc).  Only blockette::blocketteResCore (blockette.F90:299-753) is called by ref_driver.F90; none of these
! procedures is reachable from it.  Argument lists mirror the call sites in blockette.F90:70-297 so the file compiles.
module wallDistance
    use constants
contains
    subroutine updateWallDistancesQuickly(nn, level, sps)
        integer(kind=intType), intent(in) :: nn, level, sps
        stop 'stub: updateWallDistancesQuickly'
    end subroutine updateWallDistancesQuickly
end module wallDistance

module surfaceIntegrations
    use constants
contains
    subroutine getSolution(famLists, funcValues)
        integer(kind=intType), dimension(:, :), intent(in) :: famLists
        real(kind=realType), dimension(:, :), intent(out) :: funcValues
        stop 'stub: getSolution'
    end subroutine getSolution
end module surfaceIntegrations

module oversetCommUtilities
    use constants
contains
    subroutine updateOversetConnectivity(level, sps)
        integer(kind=intType), intent(in) :: level, sps
        stop 'stub: updateOversetConnectivity'
    end subroutine updateOversetConnectivity
end module oversetCommUtilities

module initializeFlow
contains
    subroutine referenceState
        stop 'stub: referenceState'
    end subroutine referenceState
end module initializeFlow

module bcdata
    use constants
contains
    subroutine setBCData(bcDataNamesIn, bcDataIn, famLists, sps, nVar, nFamMax)
        character, dimension(:, :), intent(in) :: bcDataNamesIn
        real(kind=realType), dimension(:), intent(in) :: bcDataIn
        integer(kind=intType), dimension(:, :) :: famLists
        integer(kind=intType), intent(in) :: sps
        integer, intent(in) :: nVar, nFamMax
        stop 'stub: setBCData'
    end subroutine setBCData
    subroutine setBCDataFineGrid(initializationPart)
        logical, intent(in) :: initializationPart
        stop 'stub: setBCDataFineGrid'
    end subroutine setBCDataFineGrid
end module bcdata
