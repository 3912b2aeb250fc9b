! Test-infrastructure stub (NOT product code) standing in for PETSc's Fortran
! include: maps PETSc handle types to plain 8-byte integers.
#define Vec integer(kind=8)
#define Mat integer(kind=8)
#define KSP integer(kind=8)
#define PC integer(kind=8)
#define IS integer(kind=8)
#define SNES integer(kind=8)
#define VecScatter integer(kind=8)
#define PetscFortranAddr integer(kind=8)
#define PetscErrorCode integer(kind=4)
#define PetscInt integer(kind=4)
#define PetscScalar real(kind=8)
#define PetscReal real(kind=8)
#define PetscBool logical
#define PetscViewer integer(kind=8)
#define MatNullSpace integer(kind=8)
#define PETSC_VERSION_GE(a,b,c) 1
#define PETSC_VERSION_LT(a,b,c) 0
#define PETSC_VERSION_MINOR 20
#define PETSC_VERSION_MAJOR 3
