/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.
 * The per-block kernels never reach MPI or PETSc; these definitions only let
 * the shared object load under RTLD_NOW (Python's ctypes).  Reaching one is a
 * harness bug, so they abort loudly. */
#include <stdio.h>
#include <stdlib.h>
#define STUB(name) void name(void) { fprintf(stderr, "oracle/_ref: unexpected call to " #name "\n"); abort(); }
STUB(mpi_abort_)
STUB(mpi_allreduce_)
STUB(mpi_irecv_)
STUB(mpi_isend_)
STUB(mpi_waitany_)
STUB(mpi_barrier_)
STUB(mpi_bcast_)
STUB(vecdestroy_)
STUB(vecscatterdestroy_)
