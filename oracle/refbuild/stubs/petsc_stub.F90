! Test-infrastructure stub (NOT product code): empty 'petsc' module. The PETSc
! object types are macro-mapped to integer(8) handles by the stub petsc.h.
module petsc
    implicit none
    integer, parameter :: INSERT_VALUES = 1, ADD_VALUES = 2, SCATTER_FORWARD = 0, SCATTER_REVERSE = 1
    integer, parameter :: PETSC_NULL_INTEGER = 0
end module petsc
