! Test-infrastructure stub (NOT product code): minimal 'mpi' module so the
! reference's modules/precision.F90 compiles without an MPI installation.
! Only the datatype handles referenced as parameters are provided; no MPI
! routine is ever called by the per-block kernels the oracle drives.
module mpi
    implicit none
    integer, parameter :: mpi_integer1 = 1, mpi_integer2 = 2, mpi_integer4 = 3, mpi_integer8 = 4
    integer, parameter :: mpi_real4 = 5, mpi_real8 = 6, mpi_real16 = 7
    integer, parameter :: mpi_complex = 8, mpi_double_complex = 9, mpi_complex16 = 9, mpi_complex32 = 10
    integer, parameter :: mpi_character = 11, mpi_logical = 12, mpi_integer = 3, mpi_double_precision = 6
    integer, parameter :: mpi_comm_world = 0, mpi_comm_self = 1, mpi_comm_null = 2
    integer, parameter :: mpi_sum = 1, mpi_max = 2, mpi_min = 3, mpi_lor = 4, mpi_land = 5, mpi_minloc = 6, mpi_maxloc = 7
    integer, parameter :: mpi_status_size = 6, mpi_any_source = -1, mpi_any_tag = -1
    integer, parameter :: mpi_source = 1, mpi_tag = 2, mpi_error = 3
    integer, parameter :: mpi_2integer = 13, mpi_2double_precision = 14, mpi_2real = 15
    integer, parameter :: mpi_undefined = -32766, mpi_success = 0, mpi_request_null = 0
    integer, parameter :: mpi_address_kind = 8, mpi_offset_kind = 8
    integer, parameter :: mpi_max_processor_name = 256
    integer, parameter :: mpi_in_place = 0
end module mpi
