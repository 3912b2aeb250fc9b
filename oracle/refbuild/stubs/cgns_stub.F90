! Test-infrastructure stub (NOT product code): the handful of CGNS Fortran
! parameters the reference's data-model modules mention.
module cgns
    implicit none
    integer, parameter :: cgsize_t = 4
    integer, parameter :: CG_Null = 0, CG_UserDefined = 1
    integer, parameter :: Null = 0, UserDefined = 1
    integer, parameter :: Kilogram = 2, Gram = 3, Slug = 4, PoundMass = 5
    integer, parameter :: Meter = 2, Centimeter = 3, Millimeter = 4, Foot = 5, Inch = 6
    integer, parameter :: Second = 2
    integer, parameter :: Kelvin = 2, Celcius = 3, Celsius = 3, Rankine = 4, Fahrenheit = 5
    integer, parameter :: Degree = 2, Radian = 3
    integer, parameter :: RealSingle = 3, RealDouble = 4, Integer = 2, Character = 5
    integer, parameter :: Structured = 2, Unstructured = 3
    integer, parameter :: Vertex = 2, CellCenter = 3
    integer, parameter :: CG_MODE_READ = 0, CG_MODE_WRITE = 1, CG_MODE_MODIFY = 2
    integer, parameter :: CG_OK = 0
end module cgns
